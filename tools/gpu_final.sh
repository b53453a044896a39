#!/bin/bash
# round-end evidence: full GPU suite, smoke, default bench line, rocprofv3 kernel stats of the same command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 500 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 600 gpurun_out/bench_final.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o dpo7b --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/bench_final_prof.log 2> $R/gpurun_out/bench_final_prof.err
find $R/gpurun_out/prof_final -name "*kernel_trace.csv" -delete
head -6 $R/gpurun_out/prof_final/dpo7b_kernel_stats.csv | cut -c1-160
