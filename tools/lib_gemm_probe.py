"""What the vendor library (torch.nn.functional.linear -> hipBLASLt / rocBLAS) reaches on the plain GEMM shapes of one U-Net evaluation, next to
v3d_gemm on the same operands.  A sizing probe only: the product path does not call it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from v3d_amd.ops import get_ops, GemmCall
torch.set_grad_enabled(False)
ops = get_ops()
dev = "cuda"
BF = torch.bfloat16


def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [(36864, 1920, 640, ""), (36864, 1280, 640, ""), (9216, 3840, 1280, ""), (9216, 2560, 1280, ""), (2304, 3840, 1280, ""),
          (36864, 640, 2560, "br"), (9216, 1280, 5120, "br"), (147456, 320, 320, "br"), (36864, 640, 640, "br"), (9216, 1280, 1280, "br"),
          (147456, 320, 320, "b"), (36864, 5120, 640, "b"), (4096, 4096, 4096, "")]
for M, N, K, ep in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(M, K, device=dev, generator=g).to(BF)
    w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(BF)
    b = torch.randn(N, device=dev, generator=g).to(BF) if "b" in ep else None
    r = torch.randn(M, N, device=dev, generator=g).to(BF) if "r" in ep else None
    if r is not None:
        lib = lambda: torch.addmm(r, x, w.t()) if b is None else torch.addmm(r, x, w.t()).add_(b)       # (library: residual through beta = 1; bias as a second pass)
        lib1 = lambda: torch.addmm(r, x, w.t())
    else:
        lib = lambda: F.linear(x, w, b)
        lib1 = lib
    out = torch.empty(M, N, device=dev, dtype=BF)
    bf = None if b is None else b.float()
    mine = lambda: ops.gemm(GemmCall(A=x, W=w.view(1, N, K), out=out, M=M, N=N, K=K, bias=bf, res1=r))
    t_lib, t_lib1, t_mine = timeit(lib), timeit(lib1), timeit(mine)
    fl = 2.0 * M * N * K
    ref = lib().float()
    err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
    print(f"M={M:6d} N={N:5d} K={K:5d} [{ep:2s}]  library {t_lib:7.1f} us ({fl / t_lib / 1e6:5.0f} TF/s; GEMM+residual only {t_lib1:7.1f} us)   v3d_gemm {t_mine:7.1f} us ({fl / t_mine / 1e6:5.0f} TF/s)  max rel diff {err:.1e}", flush=True)
