#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01 -o dpo7b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/bench_prof.log 2> $R/gpurun_out/bench_prof.err
tail -1 $R/gpurun_out/bench_prof.log | cut -c1-600
ls -la $R/gpurun_out/prof_r01 | head; find $R/gpurun_out/prof_r01 -name "*stats*" | head
# keep only the small summaries (the raw trace can be large)
find $R/gpurun_out/prof_r01 -name "*kernel_trace.csv" -size +20M -delete
