#!/bin/bash
# round 2, call O: persistent walk with the epilogue riding in the peeled last K-tile: correctness, lab, ksweep, bench A/B (persist on/off, fuse on/off)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_bench_geometry_gpu.py tests/test_twin_gpu.py -m gpu -q -x -k "not decoder_layer" 2>&1 | tail -3
for ps in 1 0; do
AA_GEMM_PERSIST=$ps AA_LAB_VARIANTS=base:0,g4:5 AA_LAB_BLASLT=0 AA_LAB_OUT=r2o_gemm_lab_p$ps.json timeout 600 python tools/bench_gemm_lab.py 2>&1 | grep -v amdgpu | python3 -c "
import sys,ast
rows=[ast.literal_eval(l) for l in sys.stdin if l.startswith('{')]
print('persist=$ps g4  ', ' '.join(f\"{r['name']}.{r['layout']}={max(r['g4_tf_0'],r['g4_tf_1']):.0f}\" for r in rows))
print('persist=$ps base', ' '.join(f\"{r['name']}.{r['layout']}={max(r['base_tf_0'],r['base_tf_1']):.0f}\" for r in rows))"
AA_GEMM_PERSIST=$ps AA_LAB_VARIANTS=g4:5 timeout 300 python tools/bench_gemm_ksweep.py 2>&1 | tail -1 | cut -c1-400
done
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2o_bench.json 2> gpurun_out/r2o_bench.err
AA_GEMM_PERSIST=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2o_bench_np.json 2>> gpurun_out/r2o_bench.err
AA_GEMM_FUSE=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2o_bench_nofuse.json 2>> gpurun_out/r2o_bench.err
python - <<'PY'
import json
for f in ('gpurun_out/r2o_bench.json', 'gpurun_out/r2o_bench_np.json', 'gpurun_out/r2o_bench_nofuse.json'):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['step_mfma']['frac_of_dense_bf16_peak'], d['roofline']['achieved'], d['config']['losses_timed_steps'][:3])
PY
