"""Sizing probe: what a row-resident projection kernel reaches at N = K = 320 (the 64 x 64 level's to_out / proj_in / proj_out linears) - the
existing v3d_ln_proj launched with N = 320 (LayerNorm included, bias only) next to v3d_gemm on the same rows with its epilogue variants."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from v3d_amd.ops import get_ops, GemmCall
from v3d_amd.engine.packing import ln_proj_pack
torch.set_grad_enabled(False)
ops = get_ops(); dev = "cuda"; BF = torch.bfloat16


def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M, C, S = 147456, 320, 4096
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(M, C, device=dev, generator=g).to(BF)
res = torch.randn(M, C, device=dev, generator=g).to(BF)
for N in (320, 640, 960):
    w = (torch.randn(N, C, device=dev, generator=g) / C ** 0.5)
    wp, bias = ln_proj_pack(w, torch.ones(C, device=dev), torch.zeros(C, device=dev))
    t = timeit(lambda: ops.ln_proj(x, 1e-5, wp, bias, N, S))
    print(f"ln_proj  M={M} N={N:4d} K={C}: {t:7.1f} us  ({2.0 * M * N * C / t / 1e6:5.0f} TF/s, {(M * C * 2 + M * N * 2) / t / 1e6:5.2f} TB/s)", flush=True)
w = (torch.randn(1, C, C, device=dev, generator=g) / C ** 0.5).to(BF)
bias = torch.randn(C, device=dev, generator=g)
add = torch.randn(36, C, device=dev, generator=g)
out = torch.empty(M, C, device=dev, dtype=BF)
for name, kw in (("[b]", dict(bias=bias)), ("[br]", dict(bias=bias, res1=res)), ("[bar]", dict(bias=bias, res1=res, add=add, add_rpg=S, add_ld=C))):
    t = timeit(lambda: ops.gemm(GemmCall(A=x, W=w, out=out, M=M, N=C, K=C, **kw)))
    nb = M * C * 2 * (3 if "r" in name else 2)
    print(f"v3d_gemm M={M} N={C} K={C} {name:6s}: {t:7.1f} us  ({2.0 * M * C * C / t / 1e6:5.0f} TF/s, {nb / t / 1e6:5.2f} TB/s)", flush=True)
