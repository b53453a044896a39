#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q --timeout 300 --no-header 2>&1 | tail -2
for B in 2; do
timeout 900 python bench.py --steps 6 --warmup 2 --pairs-per-gpu $B --no-cpu-baseline > gpurun_out/bench_B$B.log 2> gpurun_out/bench_B$B.err; tail -1 gpurun_out/bench_B$B.log | cut -c1-2400; tail -3 gpurun_out/bench_B$B.err
done
