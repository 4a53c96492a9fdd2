#!/bin/bash
# Timing ablations of v3d_ff_fused on the GPU box: rebuild ff.o with -DFF_ABL=<bits> and print the tick timeline (results are garbage).
# usage: tools/ff_ablate.sh 1 2 4 ...
cd "$(dirname "$0")/.."
cp v3d_amd/lib/ff.o /tmp/ff.o.keep; cp v3d_amd/lib/libv3d_hip.so /tmp/lib.keep
for a in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffast-math -fno-finite-math-only -fno-slp-vectorize -DFF_ABL=$a $FF_EXTRA -c v3d_amd/csrc/ff.hip -o v3d_amd/lib/ff.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC v3d_amd/lib/*.o -o v3d_amd/lib/libv3d_hip.so || exit 1
  echo "== FF_ABL=$a"
  timeout 120 python tools/ff_bench.py 2>&1 | grep "fused" | sed "s/^/plain /"
  V3D_FF_TIMELINE=1 timeout 120 python tools/ff_bench.py 2>&1 | grep "wave 0\|fused\|tick period\|stamp offsets"
done
cp /tmp/ff.o.keep v3d_amd/lib/ff.o; cp /tmp/lib.keep v3d_amd/lib/libv3d_hip.so
