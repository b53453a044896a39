#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_attention_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 300 --no-header > gpurun_out/tests_attn.log 2>&1
tail -5 gpurun_out/tests_attn.log
timeout 300 python tools/bench_kernels.py --no-gemm 2>&1 | grep attn
