#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
for B in 2 4; do
timeout 900 python bench.py --steps 6 --warmup 2 --pairs-per-gpu $B --no-cpu-baseline > gpurun_out/bench_B$B.log 2> gpurun_out/bench_B$B.err; tail -1 gpurun_out/bench_B$B.log | cut -c1-2400; tail -3 gpurun_out/bench_B$B.err
done
