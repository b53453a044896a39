#!/bin/bash
# round 2, call A: new parity tests at the benchmarked geometry + launch / decode / PTX tests, the clean bench line, and its kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
timeout 1200 python -m pytest tests/test_bench_geometry_gpu.py tests/test_bench_launch.py tests/test_decode_gpu.py tests/test_zz_ptx_gpu.py tests/test_ppo_gpu.py -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/r2a_tests.log
timeout 600 python bench.py > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
tail -c 3000 gpurun_out/r2a_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02a -o dpo7b --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-events > $R/gpurun_out/r2a_bench_prof.log 2> $R/gpurun_out/r2a_bench_prof.err
tail -1 $R/gpurun_out/r2a_bench_prof.log | cut -c1-400
find $R/gpurun_out/prof_r02a -name "*kernel_trace.csv" -size +20M -delete
find $R/gpurun_out/prof_r02a -name "*stats*" | head
