#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01c -o dpo7b --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-events > $R/gpurun_out/bench_prof.log 2> $R/gpurun_out/bench_prof.err
tail -1 $R/gpurun_out/bench_prof.log | cut -c1-600
ls -la $R/gpurun_out/prof_r01c | head; find $R/gpurun_out/prof_r01c -name "*stats*" | head
# keep only the small summaries (the raw trace can be large)
find $R/gpurun_out/prof_r01c -name "*kernel_trace.csv" -size +20M -delete
