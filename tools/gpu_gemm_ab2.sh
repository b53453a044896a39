#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 300 --no-header 2>&1 | tail -3
timeout 600 python tools/bench_gemm_ab.py 2>&1 | tail -12
timeout 900 python bench.py --steps 6 --warmup 2 --pairs-per-gpu 2 --no-cpu-baseline > gpurun_out/bench_B2.log 2> gpurun_out/bench_B2.err; tail -1 gpurun_out/bench_B2.log | cut -c1-1500
