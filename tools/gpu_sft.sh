#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 200 python -m pytest tests/test_sft_gpu.py -m gpu -q -x 2>&1 | tail -12
