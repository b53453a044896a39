#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
timeout 900 python bench.py --steps 6 --warmup 2 > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
