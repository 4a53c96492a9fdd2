#!/bin/bash
# Same-box A/B of the v3d_gemm tile walk (V3D_GEMM_GROUPM) and tap order (V3D_GEMM_TAPINNER): micro-benchmarks at the V3D_512 shapes, one
# process per setting (the knobs are read once per process).   gpurun -- 'bash tools/gemm_ab.sh'   -> gpurun_out/gemm_ab_<tag>.log
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out
run() { tag=$1; shift; env "$@" python $R/tools/gpu_check.py --no-check --only=lin_L1,lin_L2,conv_,convt_,vae_conv > $R/gpurun_out/gemm_ab_$tag.log 2>&1; }
run base V3D_GEMM_GROUPM=0 V3D_GEMM_TAPINNER=0
run g4 V3D_GEMM_GROUPM=4 V3D_GEMM_TAPINNER=0
run g8 V3D_GEMM_GROUPM=8 V3D_GEMM_TAPINNER=0
run g8t V3D_GEMM_GROUPM=8 V3D_GEMM_TAPINNER=1
run t V3D_GEMM_GROUPM=0 V3D_GEMM_TAPINNER=1
python - <<PY
import re, glob, os
R = "$R"
tags = ["base", "g4", "g8", "g8t", "t"]
rows = {}
for t in tags:
    for ln in open(os.path.join(R, "gpurun_out", f"gemm_ab_{t}.log")):
        m = re.match(r"(\S+)\s+M=.*?([\d.]+) ms\s+([\d.]+) TF/s", ln)
        if m:
            rows.setdefault(m.group(1), {})[t] = float(m.group(3))
print(f"{'case':26s}" + "".join(f"{t:>9s}" for t in tags) + "   (TF/s)")
for k, v in rows.items():
    print(f"{k:26s}" + "".join(f"{v.get(t, float('nan')):9.0f}" for t in tags))
PY
