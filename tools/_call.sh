cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_headline_parity_gpu.py -m gpu -q -k "reference_modules" --durations=8 ) > gpurun_out/r05h_pytest_refmodules.log 2>&1
tail -30 gpurun_out/r05h_pytest_refmodules.log
grep -E "parity\]" gpurun_out/r05h_pytest_refmodules.log | head
