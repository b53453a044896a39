#!/bin/bash
# Round-3 GPU driver: `gpurun -- bash tools/gpu_r3.sh <stage> [<stage> ...]`; every stage writes under gpurun_out/ (merged back).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
for stage in "$@"; do
  echo "=== stage $stage  $(date +%T)"
  case $stage in
    tests)       # the whole -m gpu suite (parity reports land in gpurun_out/parity)
      timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/r03_pytest.log 2>&1; tail -15 gpurun_out/r03_pytest.log ;;
    tests_new)   # only this round's new / changed tests
      timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_twin_gpu.py tests/test_secondary_geometry_gpu.py tests/test_qwen2vl_gpu.py tests/test_bench_geometry_gpu.py -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/r03_pytest_new.log 2>&1; tail -25 gpurun_out/r03_pytest_new.log ;;
    lab)         # gemm4 (16x16x32) vs gemm5 (32x32x16), 12 hot shapes, hipBLASLt as a yardstick
      AA_LAB_VARIANTS=g4:5,g5:5:-1:1 AA_LAB_OUT=r03_gemm_lab_g4_g5.json timeout 600 python tools/bench_gemm_lab.py > gpurun_out/r03_gemm_lab.log 2>&1; tail -14 gpurun_out/r03_gemm_lab.log ;;
    labpmc)      # LDS bank conflicts / MFMA busy of both kernels
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r03_pmc_lds && timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/r03_pmc_lds -o p -- python $R/tools/gemm_pmc_probe.py > $R/gpurun_out/r03_pmc_lds.log 2>&1 )
      python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r03_pmc_lds/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('(anonymous namespace)::', '')).replace('void ', '')
        if 'gemm' in k:
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/r03_pmc_lds_summary.txt', 'w') as o:
    for k, v in sorted(agg.items()):
        line = k + ': ' + ', '.join(f'{c} {sum(x) / len(x):.4g}' for c, x in sorted(v.items()))
        print(line); o.write(line + '\n')
PY
      find gpurun_out/r03_pmc_lds -name "*.csv" -size +2M -delete ;;
    bench)       # the headline line exactly as the driver runs it (in-run PMC traffic when rocprofv3 is there)
      timeout 1500 python bench.py --steps 8 --warmup 2 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; tail -c 1500 gpurun_out/r03_bench.json; tail -5 gpurun_out/r03_bench.err ;;
    bench_m32)   # same with the 32x32x16 kernels for the plain / residual GEMMs
      AA_GEMM_MFMA32=1 timeout 900 python bench.py --steps 6 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > gpurun_out/r03_bench_m32.json 2> gpurun_out/r03_bench_m32.err; tail -c 600 gpurun_out/r03_bench_m32.json
      timeout 900 python bench.py --steps 6 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > gpurun_out/r03_bench_m16.json 2> gpurun_out/r03_bench_m16.err; tail -c 600 gpurun_out/r03_bench_m16.json ;;
    prof)        # rocprofv3 kernel trace + stats of the bench command
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r03_prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof -o p -- python $R/bench.py --steps 3 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > $R/gpurun_out/r03_bench_under_rocprof.json 2> $R/gpurun_out/r03_prof.err )
      f=$(find gpurun_out/r03_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r03_dpo7b_kernel_stats.csv && head -25 "$f"
      find gpurun_out/r03_prof -name "*kernel_trace.csv" -delete ;;
    power)       # gemm4 vs gemm5 under sustained load: TFLOP/s, shader clock, package power
      timeout 300 python tools/gemm_power_ab.py > gpurun_out/r03_gemm_power_ab.log 2>&1; tail -13 gpurun_out/r03_gemm_power_ab.log ;;
    tests_fix)   # the tests changed after the first hardware run
      timeout 1200 python -m pytest tests/test_secondary_geometry_gpu.py tests/test_qwen2vl_gpu.py -m gpu -q --maxfail=40 -p no:cacheprovider -k "twin or sparse_block or four_engines or rm_trainer" > gpurun_out/r03_pytest_fix.log 2>&1; tail -8 gpurun_out/r03_pytest_fix.log ;;
    secondary)   # the other backbones' DPO steps and the decode micro-bench on this round's kernels
      timeout 400 python tools/bench_qwen2vl.py > gpurun_out/r03_bench_qwen2vl_7b_dpo.json 2> gpurun_out/r03_bench_qwen2vl.err; tail -c 400 gpurun_out/r03_bench_qwen2vl_7b_dpo.json
      timeout 400 python tools/bench_qwen2audio.py > gpurun_out/r03_bench_qwen2audio_7b_dpo.json 2> gpurun_out/r03_bench_qwen2audio.err; tail -c 400 gpurun_out/r03_bench_qwen2audio_7b_dpo.json
      timeout 400 python tools/bench_qwen3moe.py > gpurun_out/r03_bench_qwen3moe_12layers_dpo.json 2> gpurun_out/r03_bench_qwen3moe.err; tail -c 400 gpurun_out/r03_bench_qwen3moe_12layers_dpo.json ;;
    ppo_smoke)
      timeout 600 python -m pytest tests/test_qwen2vl_gpu.py -m gpu -q -p no:cacheprovider -k four_engines > gpurun_out/r03_pytest_ppo_smoke.log 2>&1; tail -5 gpurun_out/r03_pytest_ppo_smoke.log ;;
    ppo4)
      timeout 900 python tools/bench_ppo.py --prompts 4 > gpurun_out/r03_bench_ppo_4prompts.json 2> gpurun_out/r03_bench_ppo4.err; tail -c 1500 gpurun_out/r03_bench_ppo_4prompts.json; tail -5 gpurun_out/r03_bench_ppo4.err ;;
    ppo)
      timeout 1500 python tools/bench_ppo.py > gpurun_out/r03_bench_ppo.json 2> gpurun_out/r03_bench_ppo.err; tail -c 2500 gpurun_out/r03_bench_ppo.json; tail -5 gpurun_out/r03_bench_ppo.err ;;
    *) echo "unknown stage $stage" ;;
  esac
done
echo "=== done $(date +%T)"
