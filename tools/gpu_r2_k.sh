#!/bin/bash
# round 2, call K: fused epilogues (rope / swiglu fwd / swiglu bwd): bit-exact tests, model-level tests, bench A/B (fuse on/off)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -k "fused or gemm4" 2>&1 | tail -12
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_bench_geometry_gpu.py tests/test_f32_gpu.py tests/test_sft_gpu.py -m gpu -q 2>&1 | tail -6
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2k_bench.json 2> gpurun_out/r2k_bench.err
AA_GEMM_FUSE=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2k_bench_nofuse.json 2>> gpurun_out/r2k_bench.err
python - <<'PY'
import json
for f in ('gpurun_out/r2k_bench.json', 'gpurun_out/r2k_bench_nofuse.json'):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['step_mfma']['frac_of_dense_bf16_peak'], d['roofline']['achieved'], d['config']['losses_timed_steps'][:4])
    except Exception as e:
        print(f, 'ERR', e); print(open('gpurun_out/r2k_bench.err').read()[-1500:])
PY
