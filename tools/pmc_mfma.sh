#!/bin/bash
# MFMA-pipe utilisation of one guided U-Net evaluation pair (tools/pmc_eval.py) from rocprofv3 PMC counters, per kernel family:
#   SQ_VALU_MFMA_BUSY_CYCLES (cycles an MFMA is executing, summed over SIMDs) / (4 SIMDs x 256 CUs x kernel duration in shader cycles).
# One --pmc pass of its own with --kernel-trace only.   gpurun -- 'tools/pmc_mfma.sh r02'
set -e
TAG=${1:-prof}
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pm
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pm -o pm -- python $R/tools/pmc_eval.py > $R/gpurun_out/${TAG}_pmc_mfma.log 2>&1 || true
cd $R
python tools/pmc_mfma.py $(find /tmp/pm -name "*.db" | head -1) | tee gpurun_out/${TAG}_pmc_mfma_busy.txt
