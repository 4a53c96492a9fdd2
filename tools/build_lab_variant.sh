#!/bin/bash
# A/B build of the EXPERIMENTS library with one lab kernel recompiled: tools/build_lab_variant.sh <tag> "<extra -D flags>" [lab file, default gemm5.hip]
# -> v3d_amd/lib_exp/libv3d_<tag>.so (objects of lib_exp/ + the recompiled lab object).  Needs `python -m v3d_amd.build --experiments` first.
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; FLAGS=$2; F=${3:-gemm5.hip}
D=$R/v3d_amd/lib_exp/$TAG
mkdir -p $D
b=${F%.hip}
SRC=$R/tools/lab/$F; [ -f $SRC ] || SRC=$R/v3d_amd/csrc/$F
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffast-math -fno-finite-math-only -Wno-unused-function -Wno-pass-failed -DV3D_EXPERIMENTS $FLAGS -I $R/v3d_amd/csrc -c $SRC -o $D/$b.o || { echo "compile failed ($TAG)"; exit 1; }
OBJS=""
for o in $R/v3d_amd/lib_exp/*.o; do
  n=$(basename $o)
  if [ -f $D/$n ]; then OBJS="$OBJS $D/$n"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $R/v3d_amd/lib_exp/libv3d_$TAG.so && echo "built lib_exp/libv3d_$TAG.so"
