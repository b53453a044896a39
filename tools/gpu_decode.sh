#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_decode_gpu.py -m gpu -q --timeout 300 --no-header > gpurun_out/tests_decode.log 2>&1
grep -E "^E  |passed|failed|Error" gpurun_out/tests_decode.log | head -30
