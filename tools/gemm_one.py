"""Run a few v3d_gemm shapes a handful of times (for rocprofv3 --pmc passes).  env SHAPES=sq,conv,ff2,geglu"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from v3d_amd.hip import HipOps
from v3d_amd.ops import GEMM_CONV3X3, GEMM_LINEAR, GemmCall
BF = torch.bfloat16
hip = HipOps()
def run(M, N, K, geglu=False, conv=None, res=False, reps=3):
    kw, taps, mode = {}, 1, GEMM_LINEAR
    a_rows = M
    if conv:
        n, H, W = conv
        M = a_rows = n * H * W
        taps, mode = 9, GEMM_CONV3X3
        kw.update(Hin=H, Win=W, Hout=H, Wout=W, stride=1, up=1)
    A = torch.randn(a_rows, K, device="cuda").to(BF)
    Wt = (torch.randn(taps, N, K, device="cuda") / (K * taps) ** 0.5).to(BF)
    o = torch.empty(M, N // 2 if geglu else N, dtype=BF, device="cuda")
    if res:
        kw["res1"] = torch.randn_like(o)
    call = GemmCall(A=A, W=Wt, out=o, M=M, N=N, K=K, bias=torch.randn(N, device="cuda"), mode=mode, geglu=geglu, **kw)
    for _ in range(reps):
        hip.gemm(call)
    torch.cuda.synchronize()
sel = os.environ.get("SHAPES", "sq,conv,ff2,geglu").split(",")
if "geglu" in sel: run(36 * 4096, 2560, 320, geglu=True)
if "ff2" in sel: run(36 * 4096, 320, 1280, res=True)
if "sq" in sel: run(4096, 4096, 4096)
if "conv" in sel: run(0, 320, 320, conv=(36, 64, 64))
