#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
touch align_anything_amd/csrc/gemm4.hip
AA_HIPCC_EXTRA=-DAA_G4_TIMING python -m align_anything_amd.build 2>&1 | tail -1
timeout 600 python tools/${G4T:-gemm4_timing.py} gpurun_out/${G4T:-gemm4_timing.py}.json 2>&1 | tail -24
