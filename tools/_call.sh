cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/lib_gemm_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05f_lib_gemm_probe.log
