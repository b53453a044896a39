#!/bin/bash
# strip-major rollout weights: kernel + generation tests, then the 7B rollout bench (default = swizzled) and the row-major A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 300 python -m pytest tests/test_decode_gpu.py tests/test_qwen2vl_gpu.py tests/test_ppo_gpu.py -m gpu -q -x 2>&1 | tail -8
AA_BENCH_DECODE_AB=1 AA_BENCH_DECODE_QUICK=1 AA_DECODE_SWIZZLE=0 timeout 200 python tools/bench_decode.py 2>&1 | grep ms_per_step | cut -c1-200
cp gpurun_out/bench_decode_quick.json gpurun_out/bench_decode_rowmajor.json
timeout 300 python tools/bench_decode.py 2>&1 | grep ms_per_step | cut -c1-330
