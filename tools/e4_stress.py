"""Stress the hand-managed epilogue's counted waits (round 6: the compiler's vmcnt(0) drains are gone, so a wrong count would now show as a rare wrong row):
each A/B case of tools/mainloop_ab.py that runs on an e4 kernel is launched N times into a poisoned output while a second stream saturates HBM with copies
(memory latencies 2-5x the quiet ones), and every result is compared bit for bit with the first.  Prints mismatching launches per case; exit code 1 on any.
  python tools/e4_stress.py [launches per case, default 300]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from v3d_amd.hip import HipOps
import mainloop_ab as ab

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = "cuda"
hip = HipOps()
only = ["lin_L0_320_bar", "lin_L0_320_r", "lin_L0_skip640_bar", "lin_L1_640_bar", "lin_L1_ff2_br", "lin_L2_1280_bar", "lin_L2_ff2_br", "plain_36864_1280_640_br", "ct_L0_320_gnin",
        "ct_L1_640_plain", "c3_L0_320_out", "c3_L1_640_out", "shard6_L0_320_bar"]
cases = ab.build_cases(hip, only)
side = torch.cuda.Stream()
big_a = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
big_b = torch.empty_like(big_a)
bad_total = 0
for name, mk, flop, oshape in cases:
    ref = torch.zeros(oshape, dtype=torch.bfloat16, device=dev)
    hip.gemm(mk(ref))
    torch.cuda.synchronize()
    bad = 0
    out = torch.empty(oshape, dtype=torch.bfloat16, device=dev)
    call = mk(out)
    for i in range(N):
        out.fill_(float("nan"))
        if i % 2 == 0:
            with torch.cuda.stream(side):           # every other launch runs against a 1-GB copy on another stream
                big_b.copy_(big_a)
        hip.gemm(call)
        if not torch.equal(out.view(torch.int16), ref.view(torch.int16)):
            bad += 1
    torch.cuda.synchronize()
    bad_total += bad
    print(f"{name:26s} {N} launches, {bad} differ from the first", flush=True)
sys.exit(1 if bad_total else 0)
