#!/bin/bash
# HBM traffic of the GEMM kernels of the bench command itself: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; never with trace
# domains other than --kernel-trace) over `python bench.py --steps 1 --warmup 1` at FULL depth (32 layers, 4 pairs), i.e. the launch mix
# bench.py times.  Writes gpurun_out/gemm_traffic.json (copied to profiles/r02_gemm_traffic.json, which bench.py reports as roofline.traffic).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $R/gpurun_out/pmc_bench_$set
  timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bench_$set -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-events > $R/gpurun_out/pmc_bench_$set.log 2>&1
  find $R/gpurun_out/pmc_bench_$set -name "*kernel_trace.csv" -delete
done
AA_TRAFFIC_CMD="tools/pmc_traffic.sh: bench.py --steps 1 --warmup 1, 32 layers, 4 pairs" python $R/tools/make_traffic_json.py $R/gpurun_out $R/gpurun_out/gemm_traffic.json
