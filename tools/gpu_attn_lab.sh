#!/bin/bash
# attention A/B on one box: parity suites on the new kernels, bit-identity + timing lab across builds, per-kernel rocprof stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
timeout 600 python -m pytest tests/test_attention_gpu.py tests/test_lmhead_gpu.py tests/test_bench_launch.py "tests/test_f32_gpu.py::test_opt125m_64_step_loss_curve_vs_reference" -m gpu -q --no-header 2>&1 | tail -6
timeout 600 python -m pytest tests/test_bench_geometry_gpu.py tests/test_model_gpu.py tests/test_qwen2vl_gpu.py tests/test_qwen2audio_gpu.py -m gpu -q --no-header 2>&1 | tail -4
AA_LAB_OUT=attn_lab.json timeout 900 python tools/attn_lab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/attn_lab.txt
cd /tmp && export TMPDIR=/tmp
for lib in ${AA_PROF_LIBS:-libaa_hip_old.so libaa_hip.so}; do
  rm -rf $R/gpurun_out/prof_attn_$lib
  AA_LAB_ONLY=bench AA_HIP_LIB=$R/align_anything_amd/$lib timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_attn_$lib -o a --output-format csv -- python $R/tools/attn_lab.py child > /dev/null 2>&1
  echo "== $lib"; grep -h "attn_" $R/gpurun_out/prof_attn_$lib/*kernel_stats.csv | awk -F, '{printf "%s calls=%s avg_ns=%s\n", $1, $2, $4}' | cut -c1-120
  find $R/gpurun_out/prof_attn_$lib -name "*kernel_trace.csv" -delete
done
# L2-miss read traffic per launch (FETCH_SIZE, KiB as reported; x2 for gfx950 per the guide) of the attention kernels, old vs new order
for lib in ${AA_PMC_LIBS:-libaa_hip_old.so libaa_hip.so}; do
  rm -rf $R/gpurun_out/pmc_attn_$lib
  AA_LAB_ONLY=bench AA_HIP_LIB=$R/align_anything_amd/$lib timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_attn_$lib -o a -- python $R/tools/attn_lab.py child > /dev/null 2>&1
  echo "== FETCH_SIZE $lib"
  python3 - "$R/gpurun_out/pmc_attn_$lib" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'attn_' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            acc[r['Kernel_Name'][:40]].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(f'{k:42s} launches {len(v):4d}  FETCH_SIZE avg {sum(v)/len(v):12.0f} KiB')
PY
  find $R/gpurun_out/pmc_attn_$lib -name "*kernel_trace.csv" -delete
done
