"""v3d_ff_fused vs the two v3d_gemm launches it replaces, at the 64x64 level (M = 147456, C = 320, hidden = 1280)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from v3d_amd.hip import HipOps
from tools.gpu_check import timeit
hip = HipOps()
BF = torch.bfloat16
M, C, H = 36 * 4096, 320, 1280
g = torch.Generator().manual_seed(0)
x = torch.randn(M, C, generator=g).to("cuda").to(BF)
w1 = (torch.randn(2 * H, C, generator=g) / math.sqrt(C)).to("cuda").to(BF)
b1 = torch.randn(2 * H, generator=g).to("cuda")
w2 = (torch.randn(C, H, generator=g) / math.sqrt(H)).to("cuda").to(BF)
b2 = torch.randn(C, generator=g).to("cuda")
res = torch.randn(M, C, generator=g).to("cuda").to(BF)
out = torch.empty(M, C, dtype=BF, device="cuda")
fl = 2.0 * M * C * (2 * H) + 2.0 * M * H * C
ms = timeit(lambda: hip.ff_fused(x, w1, b1, w2, b2, out, res1=res), iters=10)
print(f"fused: {ms * 1e3:.1f} us  ({fl / ms / 1e9:.0f} TF/s)")
def two():
    f = hip.linear(x, w1, b1, geglu=True)
    hip.linear(f, w2, b2, res1=res, out=out)
ms2 = timeit(two, iters=10)
print(f"two GEMMs: {ms2 * 1e3:.1f} us  ({fl / ms2 / 1e9:.0f} TF/s)")

if os.environ.get("V3D_FF_TIMELINE"):
    import ctypes, numpy as np
    buf = (ctypes.c_ulonglong * (4 * 16 * 8 + 4 * 4 * 8))()
    assert hip.lib.v3d_debug_ff_timeline(buf) == 0
    t = (np.array(buf[:], dtype=np.uint64) & np.uint64(0x7fffffffffffffff)).astype(np.int64)
    tb = t[512:].reshape(4, 4, 8)
    t = t[:512].reshape(4, 16, 8)
    print("tick period by stamp (wave 0):", [int(np.diff(t[0, 2:14, k]).mean()) for k in range(8)])
    print("stamp offsets from tick start (wave 0, mean):", [int((t[0, 2:14, k] - t[0, 2:14, 0]).mean()) for k in range(8)])
    order = [0, 4, 5, 6, 1, 2, 3]
    names = ["start", "loads_issued", "slot19", "slot39", "slots_done", "vmcnt0", "barrier"]
    for w in range(4):
        tt = t[w, 2:14][:, order]
        d = np.concatenate([np.diff(tt, axis=1)[:-1], (tt[1:, 0] - tt[:-1, 6])[:, None]], axis=1)
        print(f"wave {w}: " + "  ".join(f"->{names[(i + 1) % 7]}={d[:, i].mean():.0f}" for i in range(7)) + f"  total={d.sum(1).mean():.0f}")
    bn = ["block_start", "tick0_done", "steady_done", "pen_done", "boundary_done", "epilogue_done"]
    for w in range(1):
        for b in range(4):
            d = np.diff(tb[w, b, :6])
            nxt = (tb[w, b + 1, 0] - tb[w, b, 5]) if b < 3 else 0
            print(f"wave {w} block {b}: " + "  ".join(f"{bn[i + 1]}={d[i]}" for i in range(5)) + f"  to_next={nxt}")
