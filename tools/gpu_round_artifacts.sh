#!/bin/bash
# the round's evidence set on one box: full GPU suite, smoke, the default bench line, the same command under rocprofv3 --kernel-trace --stats,
# and the PMC traffic passes of the same command (tools/pmc_traffic.sh).  Outputs land in gpurun_out/ and are copied to profiles/ by hand.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --no-header > gpurun_out/art_tests_full.log 2>&1; grep -E "passed|failed|error" gpurun_out/art_tests_full.log | tail -3 | tee gpurun_out/art_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/art_bench.json 2> gpurun_out/art_bench.err
tail -c 1500 gpurun_out/art_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_art -o dpo7b --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-events > $R/gpurun_out/art_bench_prof.log 2> $R/gpurun_out/art_bench_prof.err
find $R/gpurun_out/prof_art -name "*kernel_trace.csv" -size +20M -delete
cd $R && bash tools/pmc_traffic.sh 2>&1 | tail -2
