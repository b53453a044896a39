#!/bin/bash
# round-2 call: generation / expert-parallel suites first, then the round's evidence set (tools/gpu_round_artifacts.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_ep_gpu.py tests/test_qwen3moe_gpu.py tests/test_decode_gpu.py tests/test_grpo_gpu.py tests/test_ppo_gpu.py -m gpu -q --no-header 2>&1 | tail -5 | tee gpurun_out/ep_tests.log
bash tools/gpu_round_artifacts.sh
