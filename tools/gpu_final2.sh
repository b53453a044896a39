#!/bin/bash
# round-end check after the rollout work: full GPU suite, smoke, 7B rollout bench (strip-major weights = default) and its row-major A/B on the same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python tools/bench_decode.py 2>&1 | grep ms_per_step | cut -c1-330
AA_DECODE_SWIZZLE=0 AA_BENCH_DECODE_AB=1 AA_BENCH_DECODE_QUICK=1 timeout 200 python tools/bench_decode.py 2>&1 | grep ms_per_step | cut -c1-250
cp gpurun_out/bench_decode_quick.json gpurun_out/bench_decode_rowmajor.json
