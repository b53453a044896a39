#!/bin/bash
# round 2, call C: gemm4 (one wave per SIMD, accumulator-file MFMAs): correctness on every layout / ragged shape, A/B on the 12 hot
# shapes, PMC on the NT o-proj shape; plus the full message of the bf16 RM fixture test
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -k "layouts_and_tiles" 2>&1 | tail -12 | tee gpurun_out/r2c_tests.log
timeout 300 python -m pytest "tests/test_ppo_gpu.py::test_rm_trainer_loss_matches_reference_fixture" -m gpu -q -x 2>&1 | grep -E "Error|assert|passed|failed" | head -20 | tee -a gpurun_out/r2c_tests.log
AA_LAB_VARIANTS=base:0,g4:5 AA_LAB_OUT=r2c_gemm_lab.json timeout 600 python tools/bench_gemm_lab.py > gpurun_out/r2c_gemm_lab.log 2>&1
tail -13 gpurun_out/r2c_gemm_lab.log | cut -c1-330
cd /tmp && export TMPDIR=/tmp
shp="nt 16384 4096 4096"; tag=g4
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  ptag=$(echo $pmc | cut -d' ' -f1)
  AA_PROBE_BLASLT=0 timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/gpurun_out/probe_pmc_${tag}_$ptag -o p -- python $R/tools/gemm_probe.py $shp 5 > /dev/null 2>&1
  find $R/gpurun_out/probe_pmc_${tag}_$ptag -name "*kernel_trace.csv" -delete
done
python3 $R/tools/pmc_summary.py $R/gpurun_out 2>/dev/null | tail -8
