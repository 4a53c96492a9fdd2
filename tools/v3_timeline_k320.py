"""Raw step timeline of the persistent 192 x 320 LINEAR kernel on the HBM-bound K = N = 320 linear (experiments library, V3D_GEMM_ABLATE bit 8):
s_memtime stamps of wave 0 (group 0) and wave 4 (group 1) of block 0 for flat steps 32 .. 63 (10 steps per tile: three tile boundaries inside).
  V3D_HIP_LIB=v3d_amd/lib_exp/libv3d_hip_exp.so python tools/v3_timeline_k320.py [bar|b]"""
import os, sys, ctypes
os.environ["V3D_GEMM_IMPL"] = "3"
os.environ["V3D_GEMM_ABLATE"] = "8"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from v3d_amd.hip import HipOps
from v3d_amd.ops import GEMM_LINEAR, GemmCall
hip = HipOps()
kind = sys.argv[1] if len(sys.argv) > 1 else "bar"
M, N, K = 2 * 147456, 320, int(os.environ.get("K", "320"))      # 1536 tiles: 6 per CU, steps 32 .. 63 lie in tiles 3 .. 6
A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(1, N, K, device="cuda") / K ** 0.5).bfloat16()
o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
kw = dict(bias=torch.randn(N, device="cuda"))
if kind == "bar":
    kw.update(add=torch.randn(M // 4096, N, device="cuda"), add_rpg=4096, add_ld=N, res1=torch.randn(M, N, device="cuda").bfloat16())
call = GemmCall(A=A, W=W, out=o, M=M, N=N, K=K, mode=GEMM_LINEAR, **kw)
for _ in range(3):
    hip.gemm(call)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); hip.gemm(call); e1.record(); torch.cuda.synchronize()
print(f"{kind}: M={M} N={N} K={K}: {e0.elapsed_time(e1) * 1e3:.1f} us for the launch ({M // 192} tiles)")
buf = (ctypes.c_ulonglong * 512)()
assert hip.lib.v3d_debug_v3_timeline(buf) == 0
t = np.array(buf[:], dtype=np.int64).reshape(2, 32, 8)
t0 = t[0, 0, 0]
names = ["top", "rd+dma", "lgkm", "g1bar", "mfma", "g0bar"]
spt = K // 32
for g in range(2):
    print(f"group {g}: step, top (ticks since step 32 of group 0), then the deltas top->rd+dma->lgkm->g1bar->mfma_done->bar_passed, and the gap to the next step's top")
    for s in range(32):
        row = t[g, s, :6]
        nxt = t[g, s + 1, 0] if s + 1 < 32 else row[5]
        mark = "  <- last step of a tile: the gap is the tile's epilogue" if (32 + s) % spt == spt - 1 else ""
        print(f"  s={32 + s:3d} top={row[0] - t0:8d}  " + " ".join(f"{row[i + 1] - row[i]:6d}" for i in range(5)) + f"   gap {nxt - row[5]:7d}{mark}")
