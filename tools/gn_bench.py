"""Time the three GroupNorm steps (v3d_groupnorm_stats / finalize / apply) on the V3D shapes (env V3D_GN_BLOCKS = block-count target of the stats grid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from v3d_amd.hip import HipOps
from v3d_amd.ops import OpsBase
from tools.gpu_check import timeit
hip = HipOps()
for (n, S, C1, C2, ips) in [(36, 4096, 320, 0, 1), (36, 4096, 320, 0, 18), (36, 4096, 320, 320, 1), (36, 1024, 640, 0, 1), (36, 1024, 640, 640, 1),
                            (36, 256, 1280, 0, 1), (36, 256, 1280, 1280, 1), (36, 64, 1280, 0, 18), (18, 512 * 512, 128, 0, 1)]:
    C = C1 + C2
    x1 = torch.randn(n * S, C1, device="cuda").bfloat16()
    x2 = torch.randn(n * S, C2, device="cuda").bfloat16() if C2 else None
    st = torch.zeros(n // ips, OpsBase.gn_nslots(ips * S, ips), 32, 2, device="cuda")
    ga, be = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    table = torch.empty(n // ips, C, 2, device="cuda")
    out = torch.empty(n * S, C, device="cuda", dtype=torch.bfloat16)
    t_s = timeit(lambda: hip.groupnorm_stats(x1, x2, st, n, S, 32, ips), iters=20)
    t_f = timeit(lambda: hip.groupnorm_finalize(st, None, ga, be, float(ips * S * (C // 32)), 1e-5, table), iters=20)
    t_a = timeit(lambda: hip.groupnorm_apply(x1, x2, table, out, n, S, ips, True), iters=20)
    gb = n * S * C * 2 / 1e9
    print(f"n{n:3d} S{S:6d} C{C1}+{C2} ips{ips:2d} slots {st.shape[1]:5d}: stats {t_s * 1e3:7.1f} us ({gb / t_s:5.0f} GB/s)  finalize {t_f * 1e3:6.1f} us  apply {t_a * 1e3:7.1f} us ({2 * gb / t_a:5.0f} GB/s)", flush=True)
