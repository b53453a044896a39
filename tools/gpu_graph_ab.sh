#!/bin/bash
# hipGraph replay of the decode position: correctness test + 7B A/B (eager launches vs one graph launch per position), same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 200 python -m pytest tests/test_decode_gpu.py -m gpu -q -x -k "hipgraph" 2>&1 | tail -3
for gr in 0 1; do
  echo "AA_BENCH_DECODE_GRAPH=$gr"
  AA_BENCH_DECODE_GRAPH=$gr AA_BENCH_DECODE_QUICK=1 AA_BENCH_DECODE_AB=1 timeout 200 python tools/bench_decode.py 2>&1 | grep ms_per_step | sed "s/.*'N': \([0-9]*\).*'graph_used': \([A-Za-z]*\).*'ms_per_step': \([0-9.]*\).*/  N=\1 graph_used=\2 ms_per_step=\3/"
  cp gpurun_out/bench_decode_quick.json gpurun_out/bench_decode_graph$gr.json
done
