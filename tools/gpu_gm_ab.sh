#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for gm in 4 0 4 0; do
  AA_GEMM_GM=$gm timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('AA_GEMM_GM=$gm', round(d['value'],4), 'pairs/s', round(d['ms_per_step'],1), 'ms  gemm', round(r['achieved'],1), 'TF')"
done
timeout 200 python -m pytest tests/test_gemm_gpu.py -m gpu -q 2>&1 | tail -1
