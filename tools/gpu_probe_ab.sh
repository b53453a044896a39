#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_ep_gpu.py -m gpu -q --no-header 2>&1 | tail -3
timeout 300 python tools/glu_probe.py 2>&1 | grep -v amdgpu.ids
for lib in libaa_hip_old.so libaa_hip.so; do
  AA_HIP_LIB=$PWD/align_anything_amd/$lib timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-gemm-events 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value'], 3), 'pairs/s', round(d['ms_per_step'], 1), 'ms/step')"
done
