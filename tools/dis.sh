#!/bin/bash
# tools/dis.sh <obj> <mangled-substring> : disassemble one kernel from v3d_amd/lib/<obj>.o (comments stripped) to stdout
T=$(mktemp -d /tmp/dis.XXXX)
objcopy -O binary --only-section=.hip_fatbin /root/repo/v3d_amd/lib/$1.o $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $T/dev.co | awk -v pat="$2" '/^[0-9a-f]+ <.*>:$/ {on = index($0, pat) > 0} on {print}' | sed 's#[ \t]*//.*##'
rm -rf $T
