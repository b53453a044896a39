#!/bin/bash
# First GPU call of a round: full GPU suite, smoke, the default bench line, the rollout bench -- one box, ~3 GPU-minutes.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_round_start.sh r02'   (outputs: gpurun_out/<tag>_*.{json,log})
TAG=${1:-rXX}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/${TAG}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 500 python bench.py > gpurun_out/${TAG}_bench_7b_default.json 2> gpurun_out/${TAG}_bench.err; tail -c 400 gpurun_out/${TAG}_bench_7b_default.json
timeout 300 python tools/bench_decode.py 2>&1 | grep ms_per_step | cut -c1-260
cp gpurun_out/bench_decode.json gpurun_out/${TAG}_bench_decode_7b.json
