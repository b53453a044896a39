#!/bin/bash
# round 2, call B: (1) new reference-pinned RM / PPO tests, (2) GEMM lab: 4-wave tile A/B on the 12 hot shapes,
# (3) what hipBLASLt runs on the two NT shapes where it beats us (kernel names + PMC: MFMA busy, LDS instr, clocks)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
timeout 600 python -m pytest tests/test_ppo_gpu.py tests/test_zz_ptx_gpu.py -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/r2b_tests.log
AA_LAB_VARIANTS=base:0,w4:4 AA_LAB_OUT=r2b_gemm_lab.json timeout 600 python tools/bench_gemm_lab.py > gpurun_out/r2b_gemm_lab.log 2>&1
tail -14 gpurun_out/r2b_gemm_lab.log | cut -c1-330
cd /tmp && export TMPDIR=/tmp
for shp in "nt 16384 4096 4096" "nt 16384 4096 11008"; do
  tag=$(echo $shp | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/probe_kt_$tag -o p --output-format csv -- python $R/tools/gemm_probe.py $shp > $R/gpurun_out/probe_kt_$tag.log 2>&1
  find $R/gpurun_out/probe_kt_$tag -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 -c "
import csv,sys
for r in csv.DictReader(open('{}')):
    n=r['Kernel_Name']
    if 'gemm' in n.lower() or 'Cijk' in n:
        print(n[:400], int(r['End_Timestamp'])-int(r['Start_Timestamp']), r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), r.get('LDS_Block_Size'), r.get('Workgroup_Size_X', r.get('Workgroup_Size')), r.get('Grid_Size_X', r.get('Grid_Size')))
" | sort | uniq -c | sort -rn | head -8
  for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
    ptag=$(echo $pmc | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/gpurun_out/probe_pmc_${tag}_$ptag -o p -- python $R/tools/gemm_probe.py $shp > /dev/null 2>&1
    find $R/gpurun_out/probe_pmc_${tag}_$ptag -name "*kernel_trace.csv" -delete
  done
done
python3 $R/tools/pmc_summary.py $R/gpurun_out 2>/dev/null | tail -40
