"""What bounds the HBM-bound K = N = 320 linears?  Timing ablations of the persistent 192 x 320 kernel (experiments library; results are garbage by design):
V3D_GEMM_ABLATE bits 1 = no output row stores, 2 = no MFMAs, 4 = no LDS-DMA at all, 2048 = activation pieces issued out of range (no fetch of A).
  for a in 0 1 2 2048 2049 4; do V3D_GEMM_ABLATE=$a V3D_HIP_LIB=v3d_amd/lib_exp/libv3d_hip_exp.so python tools/k320_ablate.py; done"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from v3d_amd.hip import HipOps
from v3d_amd.ops import GEMM_LINEAR, GemmCall
hip = HipOps()
M, N, K = 147456, 320, 320
A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(1, N, K, device="cuda") / K ** 0.5).bfloat16()
bias = torch.randn(N, device="cuda")
add = torch.randn(M // 4096, N, device="cuda")
res = torch.randn(M, N, device="cuda").bfloat16()
o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
out = []
for name, kw in (("b", dict(bias=bias)), ("bar", dict(bias=bias, add=add, add_rpg=4096, add_ld=N, res1=res))):
    call = GemmCall(A=A, W=W, out=o, M=M, N=N, K=K, mode=GEMM_LINEAR, **kw)
    for _ in range(5): hip.gemm(call)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): hip.gemm(call)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 8 * 1e3)
    out.append(f"[{name}] {sorted(ts)[2]:6.1f} us")
print(f"V3D_GEMM_ABLATE={os.environ.get('V3D_GEMM_ABLATE', '0'):>5s}  " + "  ".join(out), flush=True)
