"""Slot-level timeline of the v3 GEMM main loop (s_memtime stamps of wave 0 / wave 4 of block 0, steps 32..63)."""
import os, sys, ctypes
os.environ["V3D_GEMM_IMPL"] = "3"
os.environ["V3D_GEMM_ABLATE"] = str(8 | int(os.environ.get("ABL", "0")))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from v3d_amd.hip import HipOps
from v3d_amd.ops import GEMM_LINEAR, GemmCall
hip = HipOps()
M = N = K = int(os.environ.get("SZ", "4096"))
A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(1, N, K, device="cuda") / K ** 0.5).bfloat16()
o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
call = GemmCall(A=A, W=W, out=o, M=M, N=N, K=K, bias=torch.randn(N, device="cuda"), mode=GEMM_LINEAR)
for _ in range(3):
    hip.gemm(call)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 512)()
assert hip.lib.v3d_debug_v3_timeline(buf) == 0
names = ["top", "rd+dma_issued", "lgkm_done", "g1bar", "mfma_issued", "g0bar", "x6", "x7"]
import numpy as np
t = np.array(buf[:], dtype=np.int64).reshape(2, 32, 8)
t0 = t[0, 0, 0]
for g in range(2):
    print(f"group {g}: per-step deltas (ticks), mean over steps 40..60")
    tt = t[g, 8:28, :6]
    d = np.concatenate([np.diff(tt, axis=1)[:-1], (tt[1:, 0] - tt[:-1, 5])[:, None]], axis=1)
    lab = [f"{names[i]}->{names[(i + 1) % 6]}" for i in range(6)]
    for i in range(6):
        print(f"   {lab[i]:28s} mean {d[:, i].mean():8.1f}  min {d[:, i].min():6d} max {d[:, i].max():6d}")
    print(f"   step total {d.sum(1).mean():.1f}")
print("group1 top - group0 top (same step):", (t[1, 8:28, 0] - t[0, 8:28, 0]).mean())
