"""v3d_ln_proj at the 64 x 64 level and at shard-rank sizes with and without the tail split (V3D_LNPROJ_SPLIT=0/1, read once per process)."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from v3d_amd.hip import HipOps
from v3d_amd.engine.packing import ln_proj_pack
hip = HipOps()
g = torch.Generator(device="cuda").manual_seed(0)
for M, S in ((147456, 4096), (24576, 4096), (36864, 4096)):
    x = (torch.randn(M, 320, device="cuda", generator=g) * 1.5 + 0.3).bfloat16()
    w = (torch.randn(960, 320, device="cuda", generator=g) / 320 ** 0.5)
    gamma, beta = torch.randn(320, device="cuda", generator=g) * 0.3 + 1, torch.randn(320, device="cuda", generator=g) * 0.3
    wp, bias = ln_proj_pack(w, gamma, beta)
    for _ in range(3): o = hip.ln_proj(x, 1e-5, wp, bias, 640, S)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): o = hip.ln_proj(x, 1e-5, wp, bias, 640, S)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"V3D_LNPROJ_SPLIT={os.environ.get('V3D_LNPROJ_SPLIT','1')} M={M}: {sorted(ts)[2]:.1f} us  checksum {float(o[0].float().sum()):.3f} {float(o[1].float().sum()):.3f}", flush=True)
