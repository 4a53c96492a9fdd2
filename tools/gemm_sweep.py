"""A/B sweep of v3d_gemm configurations (env V3D_GEMM_IMPL / V3D_GEMM_CFG, read once per process) on V3D shapes.
Usage: for c in 0 1 2 3 4; do V3D_GEMM_CFG=$c python tools/gemm_sweep.py; done"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from v3d_amd.hip import HipOps  # noqa: E402
from v3d_amd.ops import GEMM_CONV3X3, GEMM_CONVT3, GEMM_LINEAR, GemmCall  # noqa: E402
from tools.gpu_check import timeit  # noqa: E402

BF = torch.bfloat16
hip = HipOps()
tag = f"impl={os.environ.get('V3D_GEMM_IMPL', '2')} cfg={os.environ.get('V3D_GEMM_CFG', '-')} abl={os.environ.get('V3D_GEMM_ABLATE', '0')}"
if "--check" in sys.argv:
    import op_cases
    from oracle.ops_emul import EmulOps
    emu = EmulOps("cuda")
    bad = 0
    for name, fn, kw, tol in op_cases.all_cases(full=True):
        if fn is not op_cases.case_gemm:
            continue
        rel, cos, ok = op_cases.run_case(hip, emu, "cuda", name, fn, kw, tol)
        bad += (not ok)
        if not ok:
            print(f"[{tag}] FAIL {name} rel={rel:.3e} cos={cos:.6f}")
    print(f"[{tag}] gemm parity failures: {bad}")
    import ctypes
    hip.lib.v3d_debug_gn_epilogue_launches.restype = ctypes.c_longlong
    print(f"[{tag}] gn epilogue launches: {hip.lib.v3d_debug_gn_epilogue_launches()}")
shapes = [
    ("lin_L0_320x320", dict(M=36 * 4096, N=320, K=320)),
    ("lin_L0_ff1_geglu", dict(M=36 * 4096, N=2560, K=320, geglu=True)),
    ("lin_L0_ff2", dict(M=36 * 4096, N=320, K=1280)),
    ("lin_L1_ff1_geglu", dict(M=36 * 1024, N=5120, K=640, geglu=True)),
    ("lin_L1_ff2", dict(M=36 * 1024, N=640, K=2560)),
    ("lin_L2_1280", dict(M=36 * 256, N=1280, K=1280)),
    ("lin_L2_ff1_geglu", dict(M=36 * 256, N=10240, K=1280, geglu=True)),
    ("lin_sq_4096", dict(M=4096, N=4096, K=4096)),
    ("lin_sq_8192", dict(M=8192, N=8192, K=8192)),
    ("conv_L0_320", dict(N=320, K=320, conv=(36, 64, 64))),
    ("conv_L1_640", dict(N=640, K=640, conv=(36, 32, 32))),
    ("conv_L2_1280", dict(N=1280, K=1280, conv=(36, 16, 16))),
    ("conv_L3_1280", dict(N=1280, K=1280, conv=(36, 8, 8))),
    ("convt_L0_320", dict(N=320, K=320, convt=(2, 18, 4096))),
    ("vae_conv_512sq_128", dict(N=128, K=128, conv=(18, 512, 512))),
]
only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
if only:
    shapes = [sh for sh in shapes if any(o in sh[0] for o in only[0].split(","))]
line = []
for name, d in shapes:
    kw = {}
    taps, mode = 1, GEMM_LINEAR
    N, K = d["N"], d["K"]
    if "conv" in d:
        n, H, W = d["conv"]
        M = a_rows = n * H * W
        taps, mode = 9, GEMM_CONV3X3
        kw.update(Hin=H, Win=W, Hout=H, Wout=W, stride=1, up=1)
    elif "convt" in d:
        B, T, S = d["convt"]
        M = a_rows = B * T * S
        taps, mode = 3, GEMM_CONVT3
        kw.update(T=T, S=S, tmin=0, tmax=T - 1)
    else:
        M = a_rows = d["M"]
    A = torch.randn(a_rows, K, device="cuda").to(BF)
    Wt = (torch.randn(taps, N, K, device="cuda") / (K * taps) ** 0.5).to(BF)
    g = d.get("geglu", False)
    o = torch.empty(M, N // 2 if g else N, dtype=BF, device="cuda")
    call = GemmCall(A=A, W=Wt, out=o, M=M, N=N, K=K, bias=torch.randn(N, device="cuda"), mode=mode, geglu=g, **kw)
    ms = timeit(lambda: hip.gemm(call), iters=20)
    tf = 2.0 * M * N * K * taps / ms / 1e9
    line.append(f"{name}={tf:.0f}")
print(f"[{tag}] " + " ".join(line), flush=True)
