#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/ (copy what should be judged into profiles/):
#   tools/profile.sh <tag>        e.g.  gpurun --timeout 1500 -- 'tools/profile.sh r01l'
# 1. kernel trace + stats of the bench command (2 samples: warm-up + timed)  -> <tag>_kernel_stats.txt
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes over tools/pmc_eval.py (2 U-Net evaluations + a calibration
#    copy of known size), corrected as MI355X_MICROARCH.md prescribes                               -> <tag>_pmc_traffic.{txt,json}
# --pmc is never combined with sys/runtime/hip/hsa traces (only --kernel-trace).
set -e
TAG=${1:-prof}
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
rm -rf /tmp/st /tmp/pf /tmp/pw
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/st -o st -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/${TAG}_prof_bench.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/st -name "*.db" | head -1) $R/gpurun_out/${TAG}_kernel_stats.txt \
  "rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 (2 samples: warm-up + timed)" | tail -30
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf -- python $R/tools/pmc_eval.py > $R/gpurun_out/${TAG}_pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pw -- python $R/tools/pmc_eval.py > $R/gpurun_out/${TAG}_pmc_w.log 2>&1
cd $R
python tools/pmc_traffic.py $(find /tmp/pf -name "*.db" | head -1) $(find /tmp/pw -name "*.db" | head -1) 2 gpurun_out/${TAG}_pmc_traffic.json | tee gpurun_out/${TAG}_pmc_traffic.txt
# which launches carry the traffic: per-shape-class measured vs algorithmic bytes (call list written by pmc_eval.py)
python tools/pmc_per_launch.py $(find /tmp/pf -name "*.db" | head -1) $(find /tmp/pw -name "*.db" | head -1) gpurun_out/pmc_eval_calls.json | tee gpurun_out/${TAG}_pmc_per_launch.txt
