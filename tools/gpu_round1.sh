#!/bin/bash
# First GPU call: probes + per-file kernel tests (separate processes so one fault does not hide the rest)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
for f in probe rl_math optim elementwise gemm attention; do
  echo "=== $f" | tee -a gpurun_out/tests.log
  timeout 600 python -m pytest tests/test_${f}_gpu.py -m gpu -q --timeout 300 --no-header 2>&1 | tail -40 | tee -a gpurun_out/tests.log
done
timeout 900 python tools/bench_kernels.py 2>&1 | tail -150 | tee gpurun_out/bench_kernels.log
