#!/bin/bash
# round 2, call N: L2 prefetch (K-tile + 4) with counted vmcnt: correctness + A/B vs the previous commit's numbers on the same box is not possible -> lab + ksweep + bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_bench_geometry_gpu.py -m gpu -q -x -k "not decoder_layer" 2>&1 | tail -3
for ps in 1 0; do
AA_GEMM_PERSIST=$ps AA_LAB_VARIANTS=base:0,g4:5 AA_LAB_BLASLT=0 AA_LAB_OUT=r2n_gemm_lab_p$ps.json timeout 600 python tools/bench_gemm_lab.py 2>&1 | grep -v amdgpu | python3 -c "
import sys,ast
rows=[ast.literal_eval(l) for l in sys.stdin if l.startswith('{')]
print('persist=$ps g4  ', ' '.join(f\"{r['name']}.{r['layout']}={max(r['g4_tf_0'],r['g4_tf_1']):.0f}\" for r in rows))
print('persist=$ps base', ' '.join(f\"{r['name']}.{r['layout']}={max(r['base_tf_0'],r['base_tf_1']):.0f}\" for r in rows))"
done
AA_LAB_VARIANTS=g4:5 timeout 300 python tools/bench_gemm_ksweep.py 2>&1 | tail -1 | cut -c1-400
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2n_bench.json 2> gpurun_out/r2n_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r2n_bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['step_mfma']['frac_of_dense_bf16_peak'], d['roofline']['achieved'], d['config']['losses_timed_steps'][:3])
PY
