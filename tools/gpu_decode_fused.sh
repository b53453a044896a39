#!/bin/bash
# decode kernels round 2: tests, rollout bench at 7B (new sampling + attention kernels; folded-prologue variant as A/B), 16-wave strips A/B, kernel profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
timeout 300 python -m pytest tests/test_decode_gpu.py tests/test_qwen3moe_gpu.py -m gpu -q -x 2>&1 | tail -15
timeout 300 python tools/bench_decode.py 2>&1 | tail -7 | cut -c1-330
cp gpurun_out/bench_decode.json gpurun_out/bench_decode_nw8.json
AA_SKINNY_NW_SMALL=16 AA_BENCH_DECODE_QUICK=1 timeout 300 python tools/bench_decode.py 2>&1 | tail -1 | cut -c1-330
cd /tmp && export TMPDIR=/tmp
AA_BENCH_DECODE_QUICK=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_decode -o decode7b --output-format csv -- python $R/tools/bench_decode.py > $R/gpurun_out/prof_decode.log 2>&1
find $R/gpurun_out/prof_decode -name "*kernel_trace.csv" -delete
head -12 $R/gpurun_out/prof_decode/decode7b_kernel_stats.csv | cut -c1-150
