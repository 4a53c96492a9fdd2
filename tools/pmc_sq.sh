#!/bin/bash
# SQ wave-state counters of one micro-benchmark case (where do the waves of a kernel spend their cycles: parked on s_waitcnt / s_barrier,
# stalled at issue, issuing): two --pmc passes with --kernel-trace only.   gpurun -- 'tools/pmc_sq.sh <lib.so> <case> [tag]'
R=$(cd "$(dirname "$0")/.." && pwd)
LIB=$1; CASE=$2; TAG=${3:-sq}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/sq1 /tmp/sq2 /tmp/sq3
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d /tmp/sq1 -o sq -- python $R/tools/mainloop_ab.py x=$LIB --only=$CASE --rounds=1 > $R/gpurun_out/${TAG}_p1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC -d /tmp/sq2 -o sq -- python $R/tools/mainloop_ab.py x=$LIB --only=$CASE --rounds=1 > $R/gpurun_out/${TAG}_p2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d /tmp/sq3 -o sq -- python $R/tools/mainloop_ab.py x=$LIB --only=$CASE --rounds=1 > $R/gpurun_out/${TAG}_p3.log 2>&1
cd $R
for d in /tmp/sq1 /tmp/sq2 /tmp/sq3; do
  db=$(find $d -name "*.db" | head -1)
  [ -n "$db" ] && python tools/pmc_report.py $db | grep -A 12 -E "conv_halo|gemm_kernel_v"
done | tee gpurun_out/${TAG}_sq.txt
