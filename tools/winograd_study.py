"""CPU study for the next round: would a Winograd F(2x2, 3x3) formulation of the 3x3 convolutions (2.25x fewer multiplies: 400 ms of the
1460 ms sample are 3x3 convolutions) hold the per-op parity bar (max rel <= 2e-2, cosine >= 0.999 against fp32/fp64) with bf16 MFMA operands?

Simulated pipeline = what a kernel would do: bf16 activations -> input transform B^T d B in fp32 -> ROUNDED TO bf16 (MFMA operand);
weights transformed G g G^T in fp32 at pack time -> bf16; per-frequency contraction over channels with fp32 accumulation (MFMA);
inverse transform A^T m A in fp32; output rounded to bf16.  Compared with the direct form (bf16 operands, fp32 accumulation) against an
fp64 convolution of the same bf16 inputs.  Activations: SiLU(GroupNorm(x)) of random inputs with per-channel offsets (what feeds the
convolutions of the U-Net); weights: the reference's default init scale (kaiming-uniform-like, 1 / sqrt(9 K)).   python tools/winograd_study.py"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
BF = torch.bfloat16
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def winograd(x_bf, w_bf, round_v=True, round_u=True):
    """x [n, K, H, W] bf16, w [N, K, 3, 3] bf16 -> [n, N, H, W] fp32 (pad 1)."""
    n, K, H, W = x_bf.shape
    N = w_bf.shape[0]
    xp = F.pad(x_bf.float(), (1, 1, 1, 1))
    # tiles of 4x4 with stride 2
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                         # [n, K, H/2, W/2, 4, 4]
    V = torch.einsum("ij,nkhwjl,ml->nkhwim", Bt.float(), t, Bt.float())
    U = torch.einsum("ij,okjl,ml->okim", G.float(), w_bf.float(), G.float())
    if round_v:
        V = V.to(BF).float()
    if round_u:
        U = U.to(BF).float()
    M = torch.einsum("nkhwim,okim->nohwim", V, U)                    # fp32 accumulation over k
    Y = torch.einsum("ij,nohwjl,ml->nohwim", At.float(), M, At.float())   # [n, N, H/2, W/2, 2, 2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(n, N, H, W)


def rel_cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).abs().max() / b.abs().max()).item(), F.cosine_similarity(a, b, dim=0).item()


def study(K, N, H, W, n=2, mean=0.5):
    x = torch.randn(n, K, H, W) + (torch.rand(1, K, 1, 1) * 2 - 1) * mean
    x = F.silu(F.group_norm(x, 32)).to(BF)
    w = ((torch.rand(N, K, 3, 3) * 2 - 1) / (9 * K) ** 0.5).to(BF)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    direct = F.conv2d(x.float(), w.float(), padding=1).to(BF)
    wino = winograd(x, w).to(BF)
    wino_f32v = winograd(x, w, round_v=False).to(BF)
    d, wn, wf = rel_cos(direct, ref), rel_cos(wino, ref), rel_cos(wino_f32v, ref)
    print(f"K={K:5d} N={N:4d} {H}x{W}:  direct rel {d[0]:.2e} cos {d[1]:.6f} | winograd (bf16 V, bf16 U) rel {wn[0]:.2e} cos {wn[1]:.6f} | "
          f"(fp32 V, bf16 U) rel {wf[0]:.2e} cos {wf[1]:.6f}")
    return wn


if __name__ == "__main__":
    worst = (0.0, 1.0)
    for K, N, H, W in ((320, 64, 32, 32), (640, 64, 32, 32), (1280, 64, 16, 16), (2560, 32, 16, 16), (960, 64, 32, 32)):
        r = study(K, N, H, W)
        worst = (max(worst[0], r[0]), min(worst[1], r[1]))
    print(f"worst Winograd F(2x2,3x3) with bf16 operands: max rel {worst[0]:.2e}, cosine {worst[1]:.6f}   (per-op bar: 2e-2 / 0.999)")
