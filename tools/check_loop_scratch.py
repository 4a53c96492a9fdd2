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
        extra += os.environ.get("EXTRA_FLAGS", "").split()       # e.g. EXTRA_FLAGS="-DE4_DEPTH=2": audit an A/B build
        subprocess.run(["hipcc", *FLAGS, *extra, "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    parts = re.split(r"\n(_Z[^\n:]+):[^\n]*\n", text)
    dirty = 0
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1]
        if "v_mfma" not in body or (pats and not any(p in name for p in pats)):
            continue
        blocks = re.split(r"\n(\.LBB[0-9_]+:[^\n]*|; %bb\.[0-9]+:[^\n]*)", body)
        # per innermost loop: MFMAs, scratch ops in HOT blocks (blocks every step runs: they hold MFMAs, the step barrier or LDS-DMA issues),
        # scratch ops in the loop's other blocks (the loaders' item / tile / tap switches: run once per item, their compiler-made vmcnt(0)
        # only makes the counted waits conservative)
        per = collections.defaultdict(lambda: [0, 0, 0])
        for j in range(1, len(blocks), 2):
            m = re.search(r"in Loop: Header=(BB[0-9_]+) Depth=(\d+)", blocks[j])
            if m:
                b = blocks[j + 1]
                e = per[(m.group(1), int(m.group(2)))]
                ns = len(re.findall(r"scratch_(?:load|store)", b))
                e[0] += len(re.findall("v_mfma", b))
                e[1 if re.search(r"v_mfma|s_barrier|buffer_load_dwordx4[^\n]*lds", b) else 2] += ns
        # LDS-DMA issues wrapped in a readfirstlane "waterfall" loop INSIDE another loop: the compiler took a scalar operand of the DMA (the k
        # offset) for lane-varying - e.g. because it came out of an integer division, which runs on the vector ALU.  (In a prologue: harmless.)
        hdr = re.split(r"\n(\.LBB[0-9_]+:(?:[^\n]*\n\s*;[^\n]*)*)", body)
        waterfalls = sum(1 for j in range(1, len(hdr), 2) if "Parent Loop" in hdr[j] and "Inner Loop Header" in hdr[j]
                         and re.search(r"v_readfirstlane[^\n]*\n(?:[^\n]*\n){0,2}?[^\n]*v_cmp_eq[^\n]*\n[^\n]*s_and_saveexec[^\n]*\n(?:[^\n]*\n){0,2}?[^\n]*buffer_load_dwordx4[^\n]*lds[^\n]*\n[^\n]*s_xor_b64 exec",
                                       hdr[j + 1][:800]))
        loops = {k: v for k, v in per.items() if v[0]}
        bad = {k: v for k, v in loops.items() if v[1]}
        if waterfalls:
            bad["waterfall"] = waterfalls
        total = len(re.findall(r"scratch_(?:load|store)", body))
        short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:70]
        print(f"{'DIRTY' if bad else 'ok   '} {short:70s} mfma loops {len(loops)}  scratch ops in hot blocks {sum(v[1] for v in loops.values())}, "
              f"in rare blocks {sum(v[2] for v in loops.values())}  (whole kernel {total})  waterfall DMA in loops {waterfalls}")
        dirty += bool(bad)
    return 1 if dirty else 0


if __name__ == "__main__":
    sys.exit(main())
