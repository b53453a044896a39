#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
for st in 0 1 0 1; do AA_GEMM_ABLATE=$st timeout 120 python tools/bench_gemm_stagger.py 2>&1 | grep ablate; done
timeout 600 python -m pytest tests/test_twin_gpu.py tests/test_dp_gpu.py -m gpu -q 2>&1 | tail -8
