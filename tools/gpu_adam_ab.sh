#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 200 python -m pytest tests/test_optim_gpu.py -m gpu -q 2>&1 | tail -2
for thin in 0 1 0 1; do
  AA_ADAM_THIN=$thin timeout 300 python bench.py --no-cpu-baseline --no-gemm-events 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('thin=$thin', round(d['value'],4), 'pairs/s', round(d['ms_per_step'],1), 'ms')"
done
