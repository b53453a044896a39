#!/bin/bash
# full -m gpu suite + smoke on one box (log: gpurun_out/suite.log)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 --no-header > gpurun_out/suite.log 2>&1; tail -8 gpurun_out/suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
