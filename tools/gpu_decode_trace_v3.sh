#!/bin/bash
# kernel trace of the v3 rollout position (strip-major weights), 7B text geometry, 4 sequences
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
cd /tmp && export TMPDIR=/tmp
AA_BENCH_DECODE_QUICK=1 timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_decode_v3 -o decode7b --output-format csv -- python $R/tools/bench_decode.py > $R/gpurun_out/prof_decode_v3.log 2>&1
find $R/gpurun_out/prof_decode_v3 -name "*kernel_trace.csv" -delete
head -8 $R/gpurun_out/prof_decode_v3/decode7b_kernel_stats.csv | cut -c1-120
