#!/bin/bash
# round 2, call E: gemm4 schedule variants (v0 proportional, v1 DMA-first) + fixed / per-K-tile cost split
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -k "layouts_and_tiles and 5" 2>&1 | tail -3
AA_LAB_VARIANTS=base:0,g4:5:0,g4v1:5:1 AA_LAB_BLASLT=0 AA_LAB_OUT=r2e_gemm_lab.json timeout 600 python tools/bench_gemm_lab.py > gpurun_out/r2e_gemm_lab.log 2>&1
tail -13 gpurun_out/r2e_gemm_lab.log | cut -c1-400
AA_LAB_VARIANTS=base:0,g4:5:0,g4v1:5:1 timeout 300 python tools/bench_gemm_ksweep.py 2>&1 | tail -4 | cut -c1-600
