#!/bin/bash
# targeted run of the tests added since the last full-suite run (MoE decode, rule reward, expert parallelism) + the suites whose kernels/engine they touched
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_qwen3moe_gpu.py tests/test_ep_gpu.py tests/test_ppo_gpu.py tests/test_decode_gpu.py tests/test_dp_gpu.py -m gpu -q -x 2>&1 | tail -40 > gpurun_out/new_tests.log
tail -40 gpurun_out/new_tests.log
