#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 300 --no-header 2>&1 | tail -3
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_B4.log 2> gpurun_out/bench_B4.err; tail -1 gpurun_out/bench_B4.log | cut -c1-1200
