#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $R/gpurun_out/pmc_bench_$set
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bench_$set -o p -- python $R/bench.py --steps 1 --warmup 1 --layers 4 --pairs-per-gpu 4 --no-cpu-baseline --no-gemm-events > $R/gpurun_out/pmc_bench_$set.log 2>&1
  find $R/gpurun_out/pmc_bench_$set -name "*kernel_trace.csv" -delete
done
python $R/tools/make_traffic_json.py $R/gpurun_out $R/gpurun_out/gemm_traffic.json
