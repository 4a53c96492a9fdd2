cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gn or groupnorm or norm" ) > gpurun_out/r05c_pytest_gn.log 2>&1
tail -3 gpurun_out/r05c_pytest_gn.log
B="python bench.py --no-cpu-baseline --no-roofline --no-parity-rollout --steps 3 --warmup 1"
for rep in 1 2 3; do
  for v in 1 0; do echo "== GN_FUSED_TABLE=$v rep $rep"; V3D_GN_FUSED_TABLE=$v timeout 300 $B 2>&1 | grep -o '"value": [0-9.]*'; done
done
