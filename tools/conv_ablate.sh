#!/bin/bash
# Timing ablations of the LDS-haloed convolution: builds conv.hip with -DCONV_ABL=<bits> (compile-time switches: no extra branches in the
# measured loop), links each into its own library next to the product objects, runs tools/conv_gn_bench.py on it.
#   build (no GPU):  bash tools/conv_ablate.sh build "0 2 4 ..."         run:  gpurun -- 'bash tools/conv_ablate.sh run [case]'
# bits: 1 no epilogue, 2 no MFMA, 4 no weight DMA, 4096 no fragment reads, 8192 no halo DMA, 16384 no normalisation chain, 32768 no per-step waits / barrier
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/v3d_amd/lib_exp
mkdir -p $D $R/gpurun_out
if [ "$1" = "build" ]; then
  for ab in $2; do
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffast-math -fno-finite-math-only -Wno-unused-function -DCONV_ABL=$ab -c $R/v3d_amd/csrc/conv.hip -o $D/conv_$ab.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $R/v3d_amd/lib/*.o | grep -v /conv.o) $D/conv_$ab.o -o $D/libv3d_abl_$ab.so && echo built $ab ) &
  done
  wait
  exit 0
fi
CASE=${2:-c3_L0_320_in}
OUT=$R/gpurun_out/conv_ablate.txt
: > $OUT
for lib in $(ls $D/libv3d_abl_*.so | sort -t_ -k3 -n); do
  ab=$(basename $lib .so | sed 's/libv3d_abl_//')
  line=$(V3D_HIP_LIB=$lib timeout 120 python $R/tools/conv_gn_bench.py --only=$CASE 2>&1 | grep "$CASE" | head -3 | sed -e 's/.*fused/fused/')
  echo "ablate=$ab  $line" | tee -a $OUT
done
