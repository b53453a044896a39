#!/bin/bash
# SQ / LDS counters of the attention kernels at the bench block (separate --pmc passes, kernel-trace only)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmca_$tag
  AA_LAB_ONLY=bench timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmca_$tag -o p -- python $R/tools/attn_lab.py child > $R/gpurun_out/pmca_$tag.log 2>&1
  find $R/gpurun_out/pmca_$tag -name "*kernel_trace.csv" -delete
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmca_$tag | grep attn | grep -v delta
done
