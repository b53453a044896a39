#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_moe -o moe --output-format csv -- python $R/tools/bench_qwen3moe.py --steps 2 --warmup 1 > $R/gpurun_out/bench_moe_prof.log 2>&1
tail -1 $R/gpurun_out/bench_moe_prof.log | cut -c1-300
find $R/gpurun_out/prof_moe -name "*kernel_trace.csv" -delete
head -22 $R/gpurun_out/prof_moe/moe_kernel_stats.csv | cut -c1-150
