#!/bin/bash
# Round-4 GPU driver: `gpurun -- bash tools/gpu_r4.sh <stage> [<stage> ...]`; every stage writes under gpurun_out/ (merged back).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
for stage in "$@"; do
  echo "=== stage $stage  $(date +%T)"
  case $stage in
    attn_check)      # attn128 kernels vs the fp32 reference and the 16x16x32 kernels, timings on the bench block
      timeout 400 python tools/attn128_check.py $ATTN_ARGS > gpurun_out/r04_attn128_check.txt 2>&1; tail -30 gpurun_out/r04_attn128_check.txt ;;
    new_tests)       # this session's new / touched GPU tests
      timeout 1200 python -m pytest tests/test_optim_gpu.py tests/test_decode_gpu.py tests/test_gemm_gpu.py tests/test_f32_gpu.py tests/test_ep_gpu.py tests/test_qwen3moe_gpu.py tests/test_attention_gpu.py -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/r04_pytest_new.log 2>&1; tail -25 gpurun_out/r04_pytest_new.log ;;
    attn_bwd)        # backward variants selected by aa_attn_set_impl (ATTN_BASE / ATTN_IMPL) against each other: bit identity + timing
      timeout 500 python tools/attn128_check.py --bwd --base ${ATTN_BASE:-1} --impl ${ATTN_IMPL:-3} --out r04_attn_bwd_x.json $ATTN_ARGS > gpurun_out/r04_attn_bwd_x.txt 2>&1; python3 - <<'PY'
import json
for c in json.load(open('gpurun_out/r04_attn_bwd_x.json')):
    print(c['case'], 'ok' if c['ok'] else 'MISMATCH', {k: v for k, v in c.items() if 'identical' in k or k.endswith('_us') or k.startswith('rel_d')})
PY
      tail -1 gpurun_out/r04_attn_bwd_x.txt ;;
    bwd_lab)         # timing-only lab builds of the backward kernels (tools/build_bwd_lab.sh): where the tile time goes
      AA_LAB_CASES=bench AA_LAB_OUT=r04_attn_bwd_lab.json AA_ATTN_LIBS=$(cd align_anything_amd && ls libaa_hip_lab_*.so | sort | tr '\n' ',' | sed 's/,$//') timeout 900 python tools/attn_lab.py 2>&1 | cut -c1-140 | tee gpurun_out/r04_attn_bwd_lab.txt ;;
    attn_variants)   # the same check per lab library in AA_ATTN_LIBS (numerics only unless ATTN_VAR_ARGS says otherwise)
      for lib in ${AA_ATTN_LIBS:-libaa_hip_thr0.so}; do
        echo "--- $lib"; AA_HIP_LIB=$R/align_anything_amd/$lib timeout 300 python tools/attn128_check.py ${ATTN_VAR_ARGS:---no-time} --out r04_attn128_$lib.json > gpurun_out/r04_attn128_$lib.txt 2>&1; tail -14 gpurun_out/r04_attn128_$lib.txt | cut -c1-400
      done ;;
    attn_pmc)
      bash tools/attn128_pmc.sh $ATTN_ARGS 2>&1 | tail -60 ;;
    attn_tests)      # the existing attention / model suites on the new kernels
      timeout 600 python -m pytest tests/test_attention_gpu.py tests/test_bench_geometry_gpu.py tests/test_model_gpu.py tests/test_twin_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r04_pytest_attn.log 2>&1; tail -8 gpurun_out/r04_pytest_attn.log ;;
    moe_tests)
      timeout 600 python -m pytest tests/test_elementwise_gpu.py tests/test_qwen3moe_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r04_pytest_moe.log 2>&1; tail -8 gpurun_out/r04_pytest_moe.log ;;
    moe_bench)
      timeout 400 python tools/bench_qwen3moe.py --steps 4 --warmup 2 > gpurun_out/r04_bench_qwen3moe.json 2> gpurun_out/r04_bench_qwen3moe.err; cat gpurun_out/r04_bench_qwen3moe.json | cut -c1-1500; tail -3 gpurun_out/r04_bench_qwen3moe.err ;;
    moe_prof)
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r04_moe_prof && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_moe_prof -o p -- python $R/tools/bench_qwen3moe.py --steps 3 --warmup 1 > $R/gpurun_out/r04_bench_qwen3moe_under_rocprof.json 2> $R/gpurun_out/r04_moe_prof.err )
      f=$(find gpurun_out/r04_moe_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04_qwen3moe_kernel_stats.csv && head -30 "$f" | cut -c1-200
      t=$(find gpurun_out/r04_moe_prof -name "*kernel_trace.csv" | head -1)
      [ -n "$t" ] && python3 - "$t" <<'PY' > gpurun_out/r04_moe_trace_summary.txt
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r['Kernel_Name'].split('(')[0][-60:]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:25]:
    s = sorted(v)
    print(f'{k:62s} n {len(v):4d} total {sum(v) / 1e3:8.2f} ms  p10 {s[len(s) // 10]:9.1f}  p50 {s[len(s) // 2]:9.1f}  p90 {s[(9 * len(s)) // 10]:9.1f}  max {s[-1]:10.1f} us')
PY
      cat gpurun_out/r04_moe_trace_summary.txt | cut -c1-200
      find gpurun_out/r04_moe_prof -name "*kernel_trace.csv" -delete; tail -3 gpurun_out/r04_moe_prof.err ;;
    tests)
      timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/r04_pytest.log 2>&1; tail -15 gpurun_out/r04_pytest.log ;;
    bench)
      timeout 1500 python bench.py --steps 8 --warmup 2 > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; tail -c 1500 gpurun_out/r04_bench.json; tail -5 gpurun_out/r04_bench.err ;;
    bench_quick)     # headline step without the PMC passes / CPU leg / batch sweep
      timeout 600 python bench.py --steps 6 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > gpurun_out/r04_bench_quick.json 2> gpurun_out/r04_bench_quick.err; python -c "import json; d=json.load(open('gpurun_out/r04_bench_quick.json')); print('ms/step', d['ms_per_step'], 'pairs/s', d['value'], 'gemm frac', d['roofline']['frac'])" || tail -5 gpurun_out/r04_bench_quick.err ;;
    bench_ab)        # headline step with AA_ATTN128 = 0 / 3, alternating, same box
      for rep in 1 2; do for v in 0 3; do
        AA_ATTN128=$v timeout 600 python bench.py --steps 6 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > gpurun_out/r04_bench_attn$v.json 2> gpurun_out/r04_bench_attn$v.err
        python -c "import json; d=json.load(open('gpurun_out/r04_bench_attn$v.json')); print('AA_ATTN128=$v rep $rep', round(d['ms_per_step'],2), round(d['value'],4))" || tail -3 gpurun_out/r04_bench_attn$v.err
      done; done ;;
    secondary)       # the secondary configs at 2 and 4 pairs per step + the MoE step
      for b in 2 4; do
        timeout 400 python tools/bench_qwen2vl.py --pairs $b --steps 3 --warmup 1 > gpurun_out/r04_bench_qwen2vl_b$b.json 2> gpurun_out/r04_bench_qwen2vl_b$b.err; cut -c1-400 gpurun_out/r04_bench_qwen2vl_b$b.json; tail -2 gpurun_out/r04_bench_qwen2vl_b$b.err
        timeout 400 python tools/bench_qwen2audio.py --pairs $b --steps 3 --warmup 1 > gpurun_out/r04_bench_qwen2audio_b$b.json 2> gpurun_out/r04_bench_qwen2audio_b$b.err; cut -c1-400 gpurun_out/r04_bench_qwen2audio_b$b.json; tail -2 gpurun_out/r04_bench_qwen2audio_b$b.err
      done
      timeout 400 python tools/bench_qwen3moe.py --steps 4 --warmup 2 > gpurun_out/r04_bench_qwen3moe_final.json 2> gpurun_out/r04_bench_qwen3moe_final.err; cut -c1-600 gpurun_out/r04_bench_qwen3moe_final.json ;;
    prof_old_fwd)    # the same profile with the 16x16x32 forward (AA_ATTN128=0): in-step kernel averages of both forwards on one box
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r04_prof0 && AA_ATTN128=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_prof0 -o p -- python $R/bench.py --steps 3 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > $R/gpurun_out/r04_bench_under_rocprof_attn0.json 2> $R/gpurun_out/r04_prof0.err )
      f=$(find gpurun_out/r04_prof0 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04_dpo7b_kernel_stats_attn0.csv && grep -i "attn" "$f" | cut -c1-160
      find gpurun_out/r04_prof0 -name "*kernel_trace.csv" -delete ;;
    fwd_xl)          # in-step A/B of the head_dim-128 forward's block order (kv heads fastest vs XCD-local, lab library libaa_hip_xl1.so): step time + kernel average under rocprofv3
      for rep in 1 2; do for lib in libaa_hip.so libaa_hip_xl1.so; do
        AA_HIP_LIB=$R/align_anything_amd/$lib timeout 600 python bench.py --steps 6 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > gpurun_out/r04_bench_$lib.json 2> gpurun_out/r04_bench_$lib.err
        python -c "import json; d=json.load(open('gpurun_out/r04_bench_$lib.json')); print('$lib rep $rep', round(d['ms_per_step'],2), round(d['value'],4))" || tail -3 gpurun_out/r04_bench_$lib.err
      done; done
      for lib in libaa_hip.so libaa_hip_xl1.so; do
        ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r04_prof_$lib && AA_HIP_LIB=$R/align_anything_amd/$lib timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_prof_$lib -o p -- python $R/bench.py --steps 3 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > /dev/null 2> $R/gpurun_out/r04_prof_$lib.err )
        f=$(find gpurun_out/r04_prof_$lib -name "*kernel_stats.csv" | head -1); echo "--- $lib"; [ -n "$f" ] && grep -i "attn128" "$f" | cut -c1-140
        find gpurun_out/r04_prof_$lib -name "*kernel_trace.csv" -delete
      done ;;
    prof)
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r04_prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_prof -o p -- python $R/bench.py --steps 3 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > $R/gpurun_out/r04_bench_under_rocprof.json 2> $R/gpurun_out/r04_prof.err )
      f=$(find gpurun_out/r04_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04_dpo7b_kernel_stats.csv && head -25 "$f" | cut -c1-220
      find gpurun_out/r04_prof -name "*kernel_trace.csv" -delete ;;
    ppo_prof)        # kernel trace of one PPO iteration (Qwen2-VL-7B geometry): per-kernel split of the DECODE window (first .. last skinny GEMM), busy vs span
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r04_ppo_prof && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_ppo_prof -o p -- python $R/tools/bench_ppo.py --iters 1 --new-tokens 128 > $R/gpurun_out/r04_bench_ppo_under_rocprof.json 2> $R/gpurun_out/r04_ppo_prof.err )
      t=$(find gpurun_out/r04_ppo_prof -name "*kernel_trace.csv" | head -1)
      [ -n "$t" ] && python3 tools/decode_trace_summary.py "$t" > gpurun_out/r04_decode_trace_summary.txt; cat gpurun_out/r04_decode_trace_summary.txt | cut -c1-200
      find gpurun_out/r04_ppo_prof -name "*kernel_trace.csv" -delete ;;
    decode_fold)     # RMSNorm folded into the strip-major copies (AA_DECODE_NORM_FOLD, default 1): tests, then the PPO iteration both ways on one box
      timeout 300 python -m pytest tests/test_decode_gpu.py tests/test_qwen2vl_gpu.py tests/test_ppo_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -8
      for f in 0 1 0 1; do
        AA_DECODE_NORM_FOLD=$f timeout 200 python tools/bench_ppo.py --iters 2 > gpurun_out/r04_bench_ppo_fold$f.json 2> gpurun_out/r04_bench_ppo_fold$f.err
        python -c "import json; d=json.load(open('gpurun_out/r04_bench_ppo_fold$f.json')); print('fold $f', round(d['decode_ms_per_position'],4), 'ms/pos', round(d['iteration_ms'],1), 'ms/iter', round(d['decode_weight_stream_frac_of_hbm_peak'],4))" || tail -3 gpurun_out/r04_bench_ppo_fold$f.err
      done ;;
    *) echo "unknown stage $stage" ;;
  esac
done
echo "=== done $(date +%T)"
