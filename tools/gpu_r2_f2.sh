#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
for v in noDMA:5:2 noRD:5:3 mfmaOnly:5:4; do
  AA_LAB_VARIANTS=g4:5:0,$v timeout 120 python tools/bench_gemm_ksweep.py 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-420
done
