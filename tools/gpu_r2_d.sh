#!/bin/bash
# round 2, call D: gemm4 with the generated hand-placed schedule (asm LDS reads / DMA): correctness, A/B on the 12 hot shapes, PMC
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -k "layouts_and_tiles" 2>&1 | tail -12 | tee gpurun_out/r2d_tests.log
AA_LAB_VARIANTS=base:0,g4:5 AA_LAB_OUT=r2d_gemm_lab.json timeout 600 python tools/bench_gemm_lab.py > gpurun_out/r2d_gemm_lab.log 2>&1
tail -13 gpurun_out/r2d_gemm_lab.log | cut -c1-330
cd /tmp && export TMPDIR=/tmp
for shp in "nt 16384 4096 4096" "tn 4096 4096 16384" "nn 16384 4096 4096"; do
tag=g4_$(echo $shp | cut -d' ' -f1)
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  ptag=$(echo $pmc | cut -d' ' -f1)
  AA_PROBE_BLASLT=0 timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/gpurun_out/probe_pmc_${tag}_$ptag -o p -- python $R/tools/gemm_probe.py $shp 5 > /dev/null 2>&1
  find $R/gpurun_out/probe_pmc_${tag}_$ptag -name "*kernel_trace.csv" -delete
done
done
python3 $R/tools/pmc_summary.py $R/gpurun_out 2>/dev/null | cut -c1-700 | tail -8
