cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 500 python tools/mainloop_ab.py base=v3d_amd/lib/libv3d_hip.so noslp=v3d_amd/lib_exp/libv3d_gc_noslp.so --rounds=5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05j_gemm_conv_noslp_ab.log
