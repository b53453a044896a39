#!/bin/bash
# timing A/B: skinny GEMV reading its weights as if stored strip-major swizzled (1 KB contiguous per wave load); results of the =1 runs are garbage, only the time counts
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 200 python -m pytest tests/test_decode_gpu.py -m gpu -q -x -k "skinny" 2>&1 | tail -2
for deep in 0 1 1 0; do
  echo "AA_SKINNY_SWZ_TIMING=$deep"
  AA_SKINNY_SWZ_TIMING=$deep AA_BENCH_DECODE_QUICK=1 AA_BENCH_DECODE_AB=1 timeout 200 python tools/bench_decode.py 2>&1 | grep ms_per_step | sed "s/.*'N': \([0-9]*\).*'ms_per_step': \([0-9.]*\).*/  N=\1 ms_per_step=\2/"
done
