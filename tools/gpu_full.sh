#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/ -x -q -m gpu --timeout 600 --no-header > gpurun_out/tests_full.log 2>&1; tail -5 gpurun_out/tests_full.log
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.log | cut -c1-3000
