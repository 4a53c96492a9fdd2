#!/bin/bash
# Regenerates the evidence of a round in one GPU call (everything lands in gpurun_out/, copy what should be judged into profiles/):
#   gpurun --timeout 3000 -- 'tools/round_end.sh r03'
# profile (kernel trace + PMC traffic + MFMA busy), the headline bench line with roofline / cpu_baseline, the other bench workloads, the GPU tests.
TAG=${1:-rnd}
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
mkdir -p gpurun_out
tools/profile.sh $TAG > gpurun_out/${TAG}_profile_sh.log 2>&1
tools/pmc_mfma.sh $TAG > /dev/null 2>&1
timeout 900 python bench.py > gpurun_out/${TAG}_bench.log 2>&1
timeout 300 python bench.py --workload scene --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_scene.log 2>&1
timeout 300 python bench.py --workload scene --fp8 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_scene_fp8.log 2>&1
timeout 300 python bench.py --inputs 4 --edm-steps 50 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_bench_inputs4_steps50.log 2>&1
timeout 300 python bench.py --graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_bench_graph.log 2>&1
timeout 400 python bench.py --shard-sim 8 > gpurun_out/${TAG}_shard_sim8.log 2>&1
timeout 400 python bench.py --shard-sim 2 > gpurun_out/${TAG}_shard_sim2.log 2>&1
timeout 200 python tools/op_times.py > gpurun_out/${TAG}_op_times.log 2>&1
timeout 200 python tools/conv_gn_bench.py > gpurun_out/${TAG}_conv_gn_bench.log 2>&1
timeout 3000 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/${TAG}_pytest_gpu.log 2>&1
cp gpurun_out/parity.json gpurun_out/${TAG}_parity.json      # (later calls start from an empty gpurun_out/ on the box: keep the full-suite record under its own name)
tail -25 gpurun_out/${TAG}_pytest_gpu.log
for f in bench bench_scene bench_scene_fp8 bench_inputs4_steps50 bench_graph; do echo "== $f"; tail -c 400 gpurun_out/${TAG}_$f.log | head -c 400; echo; done
