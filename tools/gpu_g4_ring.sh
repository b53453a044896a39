#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -12
timeout 900 python -m pytest tests/test_bench_geometry_gpu.py -x -q 2>&1 | tail -5
AA_LAB_VARIANTS=g4:5 timeout 600 python tools/bench_gemm_lab.py > gpurun_out/gemm_lab_ring.json 2> gpurun_out/gemm_lab_ring.err; grep -o "'name': '[a-z_]*', 'layout': '[a-z]*'\|'g4_tf_[01]': [0-9.]*\|'hipblaslt_tf': [0-9.]*" gpurun_out/gemm_lab_ring.json | paste -sd' ' | sed "s/'name'/\n'name'/g"
timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_ring.json 2> gpurun_out/bench_ring.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_ring.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
grep -h "losses" gpurun_out/bench_ring.err | head -3
