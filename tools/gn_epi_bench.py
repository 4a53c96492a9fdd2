"""GroupNorm statistics in the producing GEMM's epilogue vs the stand-alone pass, isolated launches at the V3D_512 level-0 / level-1 shapes.
   python tools/gn_epi_bench.py        (V3D_GEMM_ABLATE=1024: epilogue without its atomics)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.gpu_check import timeit
from v3d_amd.hip import HipOps
from v3d_amd.ops import GEMM_CONV3X3, GEMM_CONVT3, GemmCall, OpsBase
hip = HipOps()
dev, BF = "cuda", torch.bfloat16
def run(name, n_img, H, W, K, N, mode, T=18):
    S, M = H * W, n_img * H * W
    A = torch.randn(M, K, device=dev).to(BF)
    taps = 9 if mode == GEMM_CONV3X3 else 3
    Wt = (torch.randn(taps, N, K, device=dev) / (K * taps) ** 0.5).to(BF)
    out = torch.empty(M, N, device=dev, dtype=BF)
    res = torch.randn(M, N, device=dev).to(BF)
    kw = dict(Hin=H, Win=W, Hout=H, Wout=W, stride=1, up=1) if mode == GEMM_CONV3X3 else dict(T=T, S=S, tmin=0, tmax=T - 1)
    base = dict(A=A, W=Wt, out=out, M=M, N=N, K=K, mode=mode, res1=res, **kw)
    for rps, tag in ((S, "2-D"), (T * S, "3-D")):
        st = torch.zeros(M // rps, OpsBase.gn_nslots(rps, rps // S), 32, 2, device=dev)
        t0 = timeit(lambda: hip.gemm(GemmCall(**base)))
        t1 = timeit(lambda: hip.gemm(GemmCall(gn_stats=st, gn_rps=rps, gn_cpg=N // 32, **base)))
        t2 = timeit(lambda: hip.groupnorm_stats(out, None, st, n_img, S, 32, rps // S))
        print(f"{name:22s} {tag}: plain {t0 * 1e3:7.1f} us   with gn epilogue {t1 * 1e3:7.1f} us (+{(t1 - t0) * 1e3:5.1f})   stand-alone stats {t2 * 1e3:6.1f} us", flush=True)
run("conv_L0_320", 36, 64, 64, 320, 320, GEMM_CONV3X3)
run("conv_L0_640to320", 36, 64, 64, 640, 320, GEMM_CONV3X3)
run("convt_L0_320", 36, 64, 64, 320, 320, GEMM_CONVT3)
run("conv_L1_640", 36, 32, 32, 640, 640, GEMM_CONV3X3)
run("conv_L2_1280", 36, 16, 16, 1280, 1280, GEMM_CONV3X3)
run("convt_L2_1280", 36, 16, 16, 1280, 1280, GEMM_CONVT3)
