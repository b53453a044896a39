#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -q --timeout 300 --no-header 2>&1 | tail -3
timeout 600 python tools/bench_gemm_ab.py 2>&1 | tail -14
