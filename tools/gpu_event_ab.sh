#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "--no-gemm-events" "--gemm-event-stride 1" "--gemm-event-stride 7" "--no-gemm-events" "--gemm-event-stride 1" "--gemm-event-stride 7"; do
  timeout 300 python bench.py --no-cpu-baseline $v 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('$v', round(d['value'],4), 'pairs/s', round(d['ms_per_step'],1), 'ms', r.get('launches'), r.get('achieved'))"
done
