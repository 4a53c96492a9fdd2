"""Slab timeline of v3d_ln_proj (instrumented build: V3D_LNPROJ_TIMELINE=1): s_memtime stamps of block 0, all waves, stream slabs 16..47.
stamps: 0 iteration top (after the bf16 rounding of the previous tile), 1 after the counted vmcnt wait, 2 after the barrier, 3 after the 10
MFMA steps (with the previous tile's staging / stores and the LDS-DMA pieces slotted in).
    V3D_LNPROJ_TIMELINE=1 [V3D_LNPROJ_CFG=0|1] python tools/lnproj_timeline.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from v3d_amd.hip import HipOps
from v3d_amd.engine.packing import ln_proj_pack
hip = HipOps()
M, N, n_rm, S = 36 * 4096, 960, 640, 4096
x = torch.randn(M, 320, device="cuda").bfloat16()
wp, pbias = ln_proj_pack((torch.randn(N, 320, device="cuda") / 18).bfloat16(), torch.ones(320, device="cuda"), torch.zeros(320, device="cuda"))
for _ in range(3):
    hip.ln_proj(x, 1e-5, wp, pbias, n_rm, S)
torch.cuda.synchronize()
NW = 8 if os.environ.get("V3D_LNPROJ_CFG") == "1" else 4
buf = (C.c_ulonglong * (8 * 32 * 8))()
hip.lib.v3d_debug_lnproj_timeline.argtypes = [C.c_void_p]
assert hip.lib.v3d_debug_lnproj_timeline(C.cast(buf, C.c_void_p)) == 0
for w in range(NW):
    print(f"wave {w}: per slab [wait, barrier, steps, ->next top]  (cycles)")
    for sl in range(31):
        b = [buf[(w * 32 + sl) * 8 + k] for k in range(4)]
        nxt = buf[(w * 32 + sl + 1) * 8]
        d = [b[1] - b[0], b[2] - b[1], b[3] - b[2], nxt - b[3]]
        print(f"  slab {16 + sl:3d}: " + " ".join(f"{v:7d}" for v in d) + f"   total {nxt - b[0]:7d}")
