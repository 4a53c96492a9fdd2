#!/bin/bash
# A/B build of the library: tools/build_variant.sh <tag> "<extra -D flags>" [file.hip ...]   (no GPU needed)
# Recompiles the listed sources of v3d_amd/csrc (default: gemm.hip conv.hip) with the extra flags, links them with the product objects of
# the other sources -> v3d_amd/lib_exp/libv3d_<tag>.so (travels to the GPU box; select with V3D_HIP_LIB or tools/mainloop_ab.py tag=path).
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; FLAGS=$2; shift; shift
FILES=${@:-gemm.hip conv.hip}
D=$R/v3d_amd/lib_exp/$TAG
mkdir -p $D
python -m v3d_amd.build > /dev/null 2>&1        # product objects up to date
pids=""
for f in $FILES; do
  b=${f%.hip}
  x=""; { [ "$f" = "ff.hip" ] || [ "$f" = "attn.hip" ]; } && x="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffast-math -fno-finite-math-only -Wno-unused-function -Wno-pass-failed $x $FLAGS -c $R/v3d_amd/csrc/$f -o $D/$b.o &
  pids="$pids $!"
done
for p in $pids; do wait $p || { echo "compile failed ($TAG)"; exit 1; }; done
OBJS=""
for o in $R/v3d_amd/lib/*.o; do
  b=$(basename $o)
  if [ -f $D/$b ]; then OBJS="$OBJS $D/$b"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $R/v3d_amd/lib_exp/libv3d_$TAG.so && echo "built lib_exp/libv3d_$TAG.so"
