#!/bin/bash
# same box: is the step time bimodal (754 vs ~795 ms) and does it follow the flags (events / cpu baseline / rocprof) or the process?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
show() { python3 -c "
import sys, json
d = json.loads(open('$1').read().strip().splitlines()[-1]); r = d.get('roofline', {})
print('$2', round(d['value'], 3), 'pairs/s', round(d['ms_per_step'], 1), 'ms/step | GEMM', round(r.get('achieved', 0)), 'TF |', ' '.join(f\"{k['tflop']}TF:{k['avg_ms']}ms\" for k in r.get('by_kind_top12', [])[:8]))"; }
rocm-smi --showmemuse --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -4
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/mode_a.json 2>/dev/null; show gpurun_out/mode_a.json "A events-stride-11   "
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --gemm-event-stride 97 > gpurun_out/mode_b.json 2>/dev/null; show gpurun_out/mode_b.json "B events-stride-97   "
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-gemm-events > gpurun_out/mode_c.json 2>/dev/null; show gpurun_out/mode_c.json "C no events          "
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gemm-event-stride 11 > gpurun_out/mode_d.json 2>/dev/null; show gpurun_out/mode_d.json "D 3 steps / 1 warmup "
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/mode_e.json 2>/dev/null; show gpurun_out/mode_e.json "E = A again          "
