#!/bin/bash
# kernel resource usage (VGPR / AGPR / spills / LDS) of one object's device code: tools/kres.sh gemm [name-filter]
set -e
T=$(mktemp -d /tmp/kres.XXXX)
objcopy -O binary --only-section=.hip_fatbin /root/repo/v3d_amd/lib/$1.o $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co > $T/notes.txt
python3 - "$T/notes.txt" "${2:-}" <<'PY'
import re, sys, subprocess
t = open(sys.argv[1]).read()
flt = sys.argv[2]
for blk in t.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g(r"\.name")
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    if flt and flt not in name:
        continue
    name = name.replace("void (anonymous namespace)::", "")
    print(f"{name[:80]:80s} agpr={blk.split()[0]:>3s} vgpr={g(r'.vgpr_count'):>3s} sgpr={g(r'.sgpr_count'):>3s} spill={g(r'.vgpr_spill_count'):>3s} scratch={g(r'.private_segment_fixed_size'):>4s} lds={g(r'.group_segment_fixed_size')}")
PY
echo "device code object: $T/dev.co"
