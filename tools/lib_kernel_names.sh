#!/bin/bash
# Which kernels does the vendor library (torch -> hipBLASLt / rocBLAS) pick for the plain V3D projection shapes?  The kernel NAME encodes
# macro tile, MFMA instruction shape, depth-U, wave-group layout, prefetch depths and the stream-K / grid choice - free design input for the
# v3d_gemm main loop.  A yardstick only: the product path never calls the library.
#   gpurun --timeout 600 -- 'tools/lib_kernel_names.sh r06'     -> gpurun_out/<tag>_lib_kernel_names.txt
TAG=${1:-r06}
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
rm -rf /tmp/ln
timeout 400 rocprofv3 --kernel-trace -d /tmp/ln -o ln -- python $R/tools/lib_gemm_probe.py > $R/gpurun_out/${TAG}_lib_gemm_probe.log 2>&1
cd $R
python - "$(find /tmp/ln -name '*.db' | head -1)" > gpurun_out/${TAG}_lib_kernel_names.txt <<'EOF'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
have = [r[1] for r in db.execute("pragma table_info(kernels)")]
want = [c for c in ("grid_x", "grid_size_x", "grid_y", "grid_size_y", "workgroup_x", "workgroup_size_x", "lds_size", "lds_block_size", "arch_vgpr_count", "accum_vgpr_count",
                    "sgpr_count", "scratch_size") if c in have]
gx = next((c for c in ("grid_x", "grid_size_x") if c in have), None)
sel = ", ".join(want)
rows = db.execute(f"select name, count(*), avg(end-start), min(end-start){', ' + sel if sel else ''} from kernels group by name{', ' + gx if gx else ''} order by 3 desc").fetchall()
print("# rocprofv3 --kernel-trace -- python tools/lib_gemm_probe.py: every distinct (kernel, grid) with its average duration")
print("# columns of the kernels view:", " ".join(have))
print("# calls  avg_us  min_us ", " ".join(want), " name")
for r in rows:
    print(f"{r[1]:5d} {r[2]/1e3:8.1f} {r[3]/1e3:8.1f}  " + " ".join(f"{v}" for v in r[4:]) + f"  {r[0]}")
EOF
cat gpurun_out/${TAG}_lib_gemm_probe.log | tail -15
grep -v "^#" gpurun_out/${TAG}_lib_kernel_names.txt | grep -i -E "Cijk|gemm|MT[0-9]" | head -40
