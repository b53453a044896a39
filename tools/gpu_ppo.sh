#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_ppo_gpu.py tests/test_rl_math_gpu.py -m gpu -q --timeout 300 --no-header > gpurun_out/tests_ppo.log 2>&1
tail -40 gpurun_out/tests_ppo.log
