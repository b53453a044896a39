#!/bin/bash
# round 2, call F: ablation of the gemm4 K-tile loop (no DMA / no LDS reads / neither): where do the cycles beyond the MFMAs go?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
AA_LAB_VARIANTS=g4:5:0,noDMA:5:2,noRD:5:3,mfmaOnly:5:4 timeout 300 python tools/bench_gemm_ksweep.py 2>&1 | tail -4 | cut -c1-600
