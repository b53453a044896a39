#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
for f in elementwise rl_math model; do
  echo "=== $f" | tee -a gpurun_out/tests2.log
  timeout 600 python -m pytest tests/test_${f}_gpu.py -m gpu -q --timeout 300 --no-header 2>&1 | tail -60 | tee -a gpurun_out/tests2.log
done
