cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_headline_parity_gpu.py tests/test_engine_gpu.py tests/test_dist_gpu.py -m gpu -q --durations=25 -k "not reference_modules" ) > gpurun_out/r05a_pytest_changed.log 2>&1
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r05a_bench.log 2>&1
timeout 200 python tools/op_times.py > gpurun_out/r05a_op_times.log 2>&1
tail -30 gpurun_out/r05a_pytest_changed.log; tail -c 1500 gpurun_out/r05a_bench.log; tail -8 gpurun_out/r05a_op_times.log
