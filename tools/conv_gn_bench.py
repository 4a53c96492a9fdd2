"""Same-process A/B of the GroupNorm+SiLU -> convolution pairs of the V3D_512 U-Net at their exact shapes:
  unfused = v3d_groupnorm_apply + v3d_gemm (implicit GEMM, v3 / v2 kernels)   vs   fused = v3d_gemm with gn_in_table (conv.hip, LDS-haloed).
Both read the same (scale, shift) table.  Usage: python tools/conv_gn_bench.py [--only=substr,...]      -> gpurun_out/conv_gn_bench.json"""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from v3d_amd.hip import HipOps  # noqa: E402
from v3d_amd.ops import GEMM_CONV3X3, GEMM_CONVT3, GemmCall, OpsBase  # noqa: E402

BF, F32 = torch.bfloat16, torch.float32


def timeit(fn, iters=8, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


def main():
    hip = HipOps()
    dev = "cuda"
    only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]
    only = only[0] if only else None
    rows = []

    def case(name, N, C1, C2=0, conv=None, convt=None, res=False, add=False, gn_out=True):
        if only is not None and not any(o in name for o in only):
            return
        K = C1 + C2
        if conv is not None:
            n_img, H, W = conv
            S, ips, taps = H * W, 1, 9
            kw = dict(mode=GEMM_CONV3X3, Hin=H, Win=W, Hout=H, Wout=W)
        else:
            B, T, S = convt
            n_img, ips, taps = B * T, T, 3
            kw = dict(mode=GEMM_CONVT3, T=T, S=S, tmin=0, tmax=T - 1)
        M = n_img * S
        x1 = torch.randn(M, C1, device=dev).to(BF)
        x2 = torch.randn(M, C2, device=dev).to(BF) if C2 else None
        ga, be = torch.ones(K, device=dev), torch.zeros(K, device=dev)
        table = hip.groupnorm_table(x1, x2, ga, be, n_img, S, eps=1e-5, imgs_per_stat=ips)
        w = (torch.randn(taps, N, K, device=dev) / (K * taps) ** 0.5).to(BF)
        kw.update(bias=torch.zeros(N, device=dev))
        if res:
            kw.update(res1=torch.randn(M, N, device=dev).to(BF))
        if add:
            kw.update(add=torch.randn(n_img, N, device=dev), add_rpg=S, add_ld=N)
        out = torch.empty(M, N, dtype=BF, device=dev)
        h = torch.empty(M, K, dtype=BF, device=dev)
        skw = {}
        if gn_out:
            rps = ips * S
            st = torch.zeros((M // rps, OpsBase.gn_nslots(rps, ips), 32, 2), dtype=F32, device=dev)
            skw = dict(gn_stats=st, gn_rps=rps, gn_cpg=N // 32)
        fused_call = GemmCall(A=x1, A2=x2, W=w, out=out, M=M, N=N, K=K, gn_in=table, gn_in_rps=ips * S, gn_in_silu=True, **kw, **skw)
        plain_call = GemmCall(A=h, W=w, out=out, M=M, N=N, K=K, **kw, **skw)
        ok = hip.gemm_gn_in_supported(fused_call)
        t_apply = timeit(lambda: hip.groupnorm_apply(x1, x2, table, h, n_img, S, ips, True))
        t_conv = timeit(lambda: hip.gemm(plain_call))
        t_fused = timeit(lambda: hip.gemm(fused_call)) if ok else float("nan")
        flop = 2.0 * M * N * K * taps
        rows.append(dict(name=name, M=M, N=N, K=K, apply_us=t_apply, conv_us=t_conv, fused_us=t_fused, conv_tflops=flop / t_conv / 1e6,
                         fused_tflops=flop / t_fused / 1e6 if ok else None))
        print(f"{name:26s} M={M:7d} N={N:5d} K={K:5d}  apply {t_apply:7.1f} + conv {t_conv:7.1f} ({flop / t_conv / 1e6:6.0f} TF/s) = {t_apply + t_conv:7.1f} us"
              f"   fused {t_fused:7.1f} us ({flop / t_fused / 1e6 if ok else 0:6.0f} TF/s)   x{(t_apply + t_conv) / t_fused if ok else 0:.2f}", flush=True)

    # [V3D] ResBlock convolutions of the U-Net at batch 36 (input / output blocks; [ba] = in_layers conv with emb add, [br] = out_layers conv with skip)
    case("c3_L0_320_bare", 320, 320, conv=(36, 64, 64), gn_out=False)        # epilogue variants of one shape: bias only
    case("c3_L0_320_gnout", 320, 320, conv=(36, 64, 64))
    case("c3_L0_320_addonly", 320, 320, conv=(36, 64, 64), add=True, gn_out=False)
    case("c3_L0_320_resonly", 320, 320, conv=(36, 64, 64), res=True, gn_out=False)
    case("c3_L0_320_in", 320, 320, conv=(36, 64, 64), add=True)
    case("c3_L0_320_out", 320, 320, conv=(36, 64, 64), res=True)
    case("c3_L0_concat640", 320, 320, 320, conv=(36, 64, 64), add=True)
    case("c3_L0_concat960", 320, 640, 320, conv=(36, 64, 64), add=True)
    case("c3_L1_640_out", 640, 640, conv=(36, 32, 32), res=True)
    case("c3_L1_concat1920", 640, 1280, 640, conv=(36, 32, 32), add=True)
    case("c3_L2_1280_out", 1280, 1280, conv=(36, 16, 16), res=True)
    case("c3_L2_concat2560", 1280, 1280, 1280, conv=(36, 16, 16), add=True)
    case("ct_L0_320", 320, 320, convt=(2, 18, 4096), res=True, gn_out=False)
    case("ct_L0_320_gnout", 320, 320, convt=(2, 18, 4096), add=True)
    case("ct_L1_640", 640, 640, convt=(2, 18, 1024), res=True, gn_out=False)
    case("ct_L2_1280", 1280, 1280, convt=(2, 18, 256), res=True, gn_out=False)
    case("ct_L3_1280", 1280, 1280, convt=(2, 18, 64), res=True, gn_out=False)         # 48 tiles: only a stream-K tail fills the chip
    case("ct_L3_1280_add", 1280, 1280, convt=(2, 18, 64), add=True, gn_out=False)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "conv_gn_bench.json"), "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
