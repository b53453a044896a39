#!/bin/bash
# SwiGLU-backward epilogue with the saved gate / up values requested ahead of their use: parity of the fused epilogues, kernel A/B, step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_bench_geometry_gpu.py -m gpu -q --no-header -k "not decoder_layer" 2>&1 | tail -3
for lib in libaa_hip_old.so libaa_hip.so; do
  echo "== $lib"; AA_PROBE_SHORT=1 AA_HIP_LIB=$PWD/align_anything_amd/$lib timeout 300 python tools/glu_probe.py 2>&1 | grep -v amdgpu.ids | grep glu_bwd
done
for lib in libaa_hip_old.so libaa_hip.so; do
  AA_HIP_LIB=$PWD/align_anything_amd/$lib timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/glu_ab_$lib.json 2>/dev/null
  python3 -c "
import json
d = json.loads(open('gpurun_out/glu_ab_$lib.json').read().strip().splitlines()[-1]); r = d['roofline']
print('$lib', round(d['value'], 3), 'pairs/s', round(d['ms_per_step'], 1), 'ms/step |', ' '.join(f\"{k['tflop']}TF/{k['algorithmic_MB']}MB:{k['avg_ms']}ms\" for k in r['by_kind_top12'][:7]))"
done
