"""Build-time guard for the persistent MFMA kernels (gemm.hip v3, conv.hip): no scratch (register spill) traffic inside a loop that issues MFMAs.
Their main loops pace the LDS-DMA ring with COUNTED s_waitcnt vmcnt(N): a scratch_load / scratch_store in the loop is a VMEM op the count does
not know about (the wait then covers one DMA piece less than it must: a race, seen as launch-to-launch differences) and the compiler guards
every reload with vmcnt(0), draining the ring (seen as 2-3x slower launches).  Spills in the cold code between the loops are fine.
usage: python tools/check_loop_scratch.py v3d_amd/csrc/conv.hip [kernel-name-substring ...]      exit code 1 if any loop is dirty"""
import collections
import os
import re
import subprocess
import sys
import tempfile

FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffast-math", "-fno-finite-math-only"]


def main():
    src, pats = sys.argv[1], sys.argv[2:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        extra = ["-fno-slp-vectorize"] if src.endswith("ff.hip") else []
        subprocess.run(["hipcc", *FLAGS, *extra, "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    parts = re.split(r"\n(_Z[^\n:]+):[^\n]*\n", text)
    dirty = 0
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1]
        if "v_mfma" not in body or (pats and not any(p in name for p in pats)):
            continue
        blocks = re.split(r"\n(\.LBB[0-9_]+:[^\n]*|; %bb\.[0-9]+:[^\n]*)", body)
        per = collections.defaultdict(lambda: [0, 0])
        for j in range(1, len(blocks), 2):
            m = re.search(r"in Loop: Header=(BB[0-9_]+) Depth=(\d+)", blocks[j])
            if m:
                per[(m.group(1), int(m.group(2)))][0] += len(re.findall("v_mfma", blocks[j + 1]))
                per[(m.group(1), int(m.group(2)))][1] += len(re.findall(r"scratch_(?:load|store)", blocks[j + 1]))
        loops = {k: v for k, v in per.items() if v[0]}
        bad = {k: v for k, v in loops.items() if v[1]}
        total = len(re.findall(r"scratch_(?:load|store)", body))
        short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:70]
        print(f"{'DIRTY' if bad else 'ok   '} {short:70s} mfma loops {len(loops)}  scratch ops in them {sum(v[1] for v in loops.values())}  (whole kernel {total})")
        dirty += bool(bad)
    return 1 if dirty else 0


if __name__ == "__main__":
    sys.exit(main())
