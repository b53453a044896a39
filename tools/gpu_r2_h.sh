#!/bin/bash
# round 2, call H: full GPU suite + bench with gemm4 as the default 256x256 kernel (A/B vs AA_GEMM_G4=0 on the same box)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r2h_tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2h_bench_g4.json 2> gpurun_out/r2h_bench.err
python - <<'PY'
import json
for f in ('gpurun_out/r2h_bench_g4.json',):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['step_mfma']['frac_of_dense_bf16_peak'], d['roofline']['achieved'], d['config']['losses_timed_steps'])
PY
AA_GEMM_G4=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2h_bench_base.json 2>> gpurun_out/r2h_bench.err
python - <<'PY'
import json
for f in ('gpurun_out/r2h_bench_base.json',):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['step_mfma']['frac_of_dense_bf16_peak'], d['roofline']['achieved'])
PY
