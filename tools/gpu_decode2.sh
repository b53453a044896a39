#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_decode_gpu.py tests/test_ppo_gpu.py -m gpu -q --timeout 300 --no-header > gpurun_out/tests_decode.log 2>&1
grep -E "^E  |passed|failed|Error" gpurun_out/tests_decode.log | head -30
timeout 600 python tools/bench_decode.py 2>&1 | tail -7
