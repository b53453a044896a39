#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bench_$set -o p -- python $R/bench.py --steps 1 --warmup 1 --layers 4 --no-cpu-baseline --no-gemm-events > $R/gpurun_out/pmc_bench_$set.log 2>&1
  find $R/gpurun_out/pmc_bench_$set -name "*kernel_trace.csv" -delete
  ls $R/gpurun_out/pmc_bench_$set | head -3
done
