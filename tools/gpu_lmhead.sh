#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_lmhead_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/lmhead_tests.log
tail -25 gpurun_out/lmhead_tests.log
timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_lmhead.json 2> gpurun_out/bench_lmhead.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_lmhead.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
AA_LMHEAD_FUSED=0 timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_lmhead0.json 2> gpurun_out/bench_lmhead0.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_lmhead0.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
grep -h "losses" gpurun_out/bench_lmhead.err gpurun_out/bench_lmhead0.err | head
