"""One-launch GroupNorm (v3d_groupnorm_small) vs statistics -> finalize -> apply at the shapes the engine gives it.  gpurun -- python tools/gn_small_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from v3d_amd.hip import HipOps
hip = HipOps(); dev = "cuda"; BF = torch.bfloat16


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, n, S, C1, C2, ips in (("2d_L3", 36, 64, 1280, 0, 1), ("2d_L3_concat", 36, 64, 1280, 1280, 1), ("3d_L3", 36, 64, 1280, 0, 18), ("2d_L2", 36, 256, 1280, 0, 1)):
    x1 = torch.randn(n * S, C1, device=dev).to(BF)
    x2 = torch.randn(n * S, C2, device=dev).to(BF) if C2 else None
    C = C1 + C2
    ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    out = torch.empty(n * S, C, dtype=BF, device=dev)
    t_small = timeit(lambda: hip.groupnorm_small(x1, x2, ga, be, out, n, S, eps=1e-5, silu=True, imgs_per_stat=ips))

    def three():
        hip.begin_evaluation(dev)
        table = hip.groupnorm_table(x1, x2, ga, be, n, S, eps=1e-5, imgs_per_stat=ips)
        hip.groupnorm_apply(x1, x2, table, out, n, S, ips, True)
    t_three = timeit(three)
    print(f"{name:14s} one launch {t_small:6.1f} us   three-step {t_three:6.1f} us")
