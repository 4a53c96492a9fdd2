"""Per-tile fixed cost of a v3d_gemm configuration: time vs K at fixed M, N (env V3D_GEMM_IMPL / _CFG / _ABLATE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from v3d_amd.hip import HipOps
from v3d_amd.ops import GEMM_LINEAR, GemmCall
from tools.gpu_check import timeit
hip = HipOps()
M, N = 36 * 4096, int(os.environ.get("N", "2560"))
GEGLU = os.environ.get("GEGLU", "1") == "1"
RES = os.environ.get("RES", "0") == "1"
out = []
for K in (32, 64, 160, 320, 640, 1280):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(1, N, K, device="cuda") / K ** 0.5).bfloat16()
    o = torch.empty(M, N // 2 if GEGLU else N, dtype=torch.bfloat16, device="cuda")
    kw = {"res1": torch.randn_like(o)} if RES else {}
    call = GemmCall(A=A, W=W, out=o, M=M, N=N, K=K, bias=torch.randn(N, device="cuda"), mode=GEMM_LINEAR, geglu=GEGLU, **kw)
    ms = timeit(lambda: hip.gemm(call), iters=10)
    out.append(f"K{K}={ms * 1e3:.0f}us")
print(f"[N={N} geglu={int(GEGLU)} res={int(RES)} impl={os.environ.get('V3D_GEMM_IMPL', '0')} cfg={os.environ.get('V3D_GEMM_CFG', '-')} abl={os.environ.get('V3D_GEMM_ABLATE', '0')}] " + " ".join(out))
