"""v3d_ff_fused: time per launch against the number of rounds of 128-row blocks on the CUs (round 6).  Classic assignment on 256 CUs: 4.00 rounds 394 us, 4.25 464,
4.50 477, 4.75 489, 5.00 508 - a partial round is paid almost in full however few CUs run it, but those CUs also run FASTER (64 active: 70 us per row block, 128: 83,
256: 100: shared L2 / clock), so cutting the tail row blocks in two along the hidden dimension (donor / owner hand-off of the fp32 accumulators, built and measured:
475.7 vs 477.4 us at 4.5 rounds for every split point 20 .. 32 of 40 slabs) buys nothing.  Not kept; this probe is."""
import os, sys, math
sys.path.insert(0, "/root/repo")
import torch
from v3d_amd.hip import HipOps
hip = HipOps()
BF = torch.bfloat16
C, H = 320, 1280
g = torch.Generator().manual_seed(0)
w1 = (torch.randn(2 * H, C, generator=g) / math.sqrt(C)).to("cuda").to(BF)
b1 = torch.randn(2 * H, generator=g).to("cuda")
w2 = (torch.randn(C, H, generator=g) / math.sqrt(H)).to("cuda").to(BF)
b2 = torch.randn(C, generator=g).to("cuda")
for nb in (256, 512, 1024, 1088, 1152, 1216, 1280):
    M = nb * 128
    x = torch.randn(M, C, generator=g).to("cuda").to(BF)
    res = torch.randn(M, C, generator=g).to("cuda").to(BF)
    out = torch.empty(M, C, dtype=BF, device="cuda")
    fn = lambda: hip.ff_fused(x, w1, b1, w2, b2, out, res1=res)
    for _ in range(3): fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    print(f"V3D_FF_SPLIT={os.environ.get('V3D_FF_SPLIT','1')} row blocks {nb:5d} ({nb/256:.2f} rounds): {sorted(ts)[2]:7.1f} us", flush=True)
