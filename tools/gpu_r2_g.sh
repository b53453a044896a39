#!/bin/bash
# round 2, call G: gemm4 after the DMA rework (running M0, no s_nop), PLAIN instantiation with 16-byte permlane-swapped stores
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_bench_geometry_gpu.py -m gpu -q -x -k "layouts_and_tiles or epilogues or hot_gemm" 2>&1 | tail -5
AA_GEMM_TILE=5 timeout 600 python -m pytest tests/test_bench_geometry_gpu.py -m gpu -q -x -k "hot_gemm" 2>&1 | tail -3
AA_LAB_VARIANTS=base:0,g4:5 AA_LAB_BLASLT=0 AA_LAB_OUT=r2g_gemm_lab.json timeout 600 python tools/bench_gemm_lab.py > gpurun_out/r2g_gemm_lab.log 2>&1
tail -13 gpurun_out/r2g_gemm_lab.log | cut -c1-300
AA_LAB_VARIANTS=base:0,g4:5 timeout 300 python tools/bench_gemm_ksweep.py 2>&1 | tail -2 | cut -c1-500
