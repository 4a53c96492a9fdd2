cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "attn_temporal" ) > gpurun_out/r05d_pytest_attn_t.log 2>&1
tail -15 gpurun_out/r05d_pytest_attn_t.log
V3D_ATTN_TEMPORAL_IMPL=1 timeout 200 python tools/attn_temporal_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05d_attn_temporal_valu.log
V3D_ATTN_TEMPORAL_IMPL=2 timeout 200 python tools/attn_temporal_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05d_attn_temporal_mfma.log
