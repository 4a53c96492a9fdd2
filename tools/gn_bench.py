"""Time v3d_groupnorm_stats / apply on the V3D shapes (env V3D_GN_BLOCKS = block-count target of the stats grid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from v3d_amd.hip import HipOps
from v3d_amd.ops import GN_SLOTS
from tools.gpu_check import timeit
hip = HipOps()
out = []
for (n, S, C1, C2, ips) in [(36, 4096, 320, 0, 1), (36, 4096, 320, 0, 18), (36, 4096, 320, 320, 1), (36, 1024, 640, 0, 1), (36, 1024, 640, 640, 1),
                            (36, 256, 1280, 0, 1), (36, 256, 1280, 1280, 1), (36, 64, 1280, 0, 18), (18, 512 * 512, 128, 0, 1)]:
    x1 = torch.randn(n * S, C1, device="cuda").bfloat16()
    x2 = torch.randn(n * S, C2, device="cuda").bfloat16() if C2 else None
    st = torch.zeros(n // ips, GN_SLOTS, 32, 2, device="cuda")
    ms = timeit(lambda: hip.groupnorm_stats(x1, x2, st, n, S, 32, ips), iters=20)
    gb = n * S * (C1 + C2) * 2 / 1e9
    out.append(f"n{n}_S{S}_C{C1}+{C2}_ips{ips}={ms * 1e3:.1f}us({gb / ms:.0f}GB/s)")
print(f"[blocks={os.environ.get('V3D_GN_BLOCKS', 'default')}] " + " ".join(out))
