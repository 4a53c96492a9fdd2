"""Which clock does the chip hold under full-occupancy MFMA load?  (round 6)  The instrumented v3d_ff_fused build stamps, at the top of every tick of block 0,
s_memtime (shader cycles) AND s_memrealtime (a constant 100 MHz counter): cycles per tick / real time per tick = the shader clock, measured INSIDE the main loop
(launch gaps, prologue and epilogue play no part).  Run with few and with all CUs active:   V3D_FF_TIMELINE=1 python tools/clock_probe.py"""
import os, sys, math, ctypes
os.environ.setdefault("V3D_FF_TIMELINE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
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
for ncu in (16, 64, 128, 256):
    nb = ncu * 4                      # four row blocks per active CU
    M = nb * 128
    x = torch.randn(M, C, generator=g).to("cuda").to(BF)
    res = torch.randn(M, C, generator=g).to("cuda").to(BF)
    out = torch.empty(M, C, dtype=BF, device="cuda")
    # (grid = min(row blocks, CUs): to keep only `ncu` CUs busy the launch must have <= ncu row blocks per round -> call with M = ncu * 128 four times)
    xs = [x[i * ncu * 128:(i + 1) * ncu * 128] for i in range(4)]
    rs = [res[i * ncu * 128:(i + 1) * ncu * 128] for i in range(4)]
    os_ = [out[i * ncu * 128:(i + 1) * ncu * 128] for i in range(4)]
    fn = lambda: [hip.ff_fused(xs[i], w1, b1, w2, b2, os_[i], res1=rs[i]) for i in range(4)]
    for _ in range(3): fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 4 * 1e3)
    wall = sorted(ts)[2]
    buf = (ctypes.c_ulonglong * (4 * 16 * 8 + 4 * 4 * 8))()
    assert hip.lib.v3d_debug_ff_timeline(buf) == 0
    t = (np.array(buf[:], dtype=np.uint64) & np.uint64(0x7fffffffffffffff)).astype(np.int64)[:512].reshape(4, 16, 8)
    tick = float(np.diff(t[0, 2:14, 0]).mean())              # shader cycles per tick (wave 0, ticks 10 .. 21 of the only row block of block 0)
    real = float(np.diff(t[0, 2:14, 7]).mean())              # 100 MHz ticks per tick
    ghz = tick / (real * 10.0) if real > 0 else float("nan")  # cycles / ns
    print(f"{ncu:4d} CUs active: {wall:7.1f} us per one-row-block launch; inside the main loop {tick:7.0f} shader cycles = {real * 10:7.0f} ns per tick -> {ghz:5.3f} GHz; "
          f"bf16 dense peak at that clock {256 * 4 * 1024 * ghz / 1e3:6.0f} TFLOP/s", flush=True)
