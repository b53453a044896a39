#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
for st in 0 20 40 80 160 0; do AA_GEMM_STAGGER=$st timeout 120 python tools/bench_gemm_stagger.py 2>&1 | grep stagger; done
timeout 600 python -m pytest tests/test_twin_gpu.py -m gpu -q 2>&1 | tail -8
cat gpurun_out/parity_bf16_vs_fp32_twin_ulps.txt 2>/dev/null | head -40
