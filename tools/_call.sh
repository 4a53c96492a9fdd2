cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/attn_ab.py base=v3d_amd/lib/libv3d_hip.so noslp=v3d_amd/lib_exp/libv3d_attn_noslp.so defer=v3d_amd/lib_exp/libv3d_attn_defer.so defer_noslp=v3d_amd/lib_exp/libv3d_attn_defer_noslp.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05i_attn_ab.log
