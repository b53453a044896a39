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
    tests_new)   # only this round's new / changed tests (the gemm4-vs-gemm5 lab / PMC / power stages were removed with the kernel: git history, commit b2c2a50)
      timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_twin_gpu.py tests/test_secondary_geometry_gpu.py tests/test_qwen2vl_gpu.py tests/test_bench_geometry_gpu.py -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/r03_pytest_new.log 2>&1; tail -25 gpurun_out/r03_pytest_new.log ;;
    bench)       # the headline line exactly as the driver runs it (in-run PMC traffic when rocprofv3 is there)
      timeout 1500 python bench.py --steps 8 --warmup 2 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; tail -c 1500 gpurun_out/r03_bench.json; tail -5 gpurun_out/r03_bench.err ;;
    bench_m32)   # same with the 32x32x16 kernels for the plain / residual GEMMs
      AA_GEMM_MFMA32=1 timeout 900 python bench.py --steps 6 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > gpurun_out/r03_bench_m32.json 2> gpurun_out/r03_bench_m32.err; tail -c 600 gpurun_out/r03_bench_m32.json
      timeout 900 python bench.py --steps 6 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > gpurun_out/r03_bench_m16.json 2> gpurun_out/r03_bench_m16.err; tail -c 600 gpurun_out/r03_bench_m16.json ;;
    prof)        # rocprofv3 kernel trace + stats of the bench command
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r03_prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof -o p -- python $R/bench.py --steps 3 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > $R/gpurun_out/r03_bench_under_rocprof.json 2> $R/gpurun_out/r03_prof.err )
      f=$(find gpurun_out/r03_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r03_dpo7b_kernel_stats.csv && head -25 "$f"
      find gpurun_out/r03_prof -name "*kernel_trace.csv" -delete ;;
    tests_fix)   # the tests changed after the first hardware run
      timeout 1200 python -m pytest tests/test_secondary_geometry_gpu.py tests/test_qwen2vl_gpu.py -m gpu -q --maxfail=40 -p no:cacheprovider -k "twin or sparse_block or four_engines or rm_trainer" > gpurun_out/r03_pytest_fix.log 2>&1; tail -8 gpurun_out/r03_pytest_fix.log ;;
    secondary)   # the other backbones' DPO steps and the decode micro-bench on this round's kernels
      timeout 400 python tools/bench_qwen2vl.py > gpurun_out/r03_bench_qwen2vl_7b_dpo.json 2> gpurun_out/r03_bench_qwen2vl.err; tail -c 400 gpurun_out/r03_bench_qwen2vl_7b_dpo.json
      timeout 400 python tools/bench_qwen2audio.py > gpurun_out/r03_bench_qwen2audio_7b_dpo.json 2> gpurun_out/r03_bench_qwen2audio.err; tail -c 400 gpurun_out/r03_bench_qwen2audio_7b_dpo.json
      timeout 400 python tools/bench_qwen3moe.py > gpurun_out/r03_bench_qwen3moe_12layers_dpo.json 2> gpurun_out/r03_bench_qwen3moe.err; tail -c 400 gpurun_out/r03_bench_qwen3moe_12layers_dpo.json ;;
    ppo_smoke)
      timeout 600 python -m pytest tests/test_qwen2vl_gpu.py -m gpu -q -p no:cacheprovider -k four_engines > gpurun_out/r03_pytest_ppo_smoke.log 2>&1; tail -5 gpurun_out/r03_pytest_ppo_smoke.log ;;
    decode_tests)
      timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_qwen2vl_gpu.py tests/test_qwen3moe_gpu.py tests/test_ppo_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r03_pytest_decode.log 2>&1; tail -6 gpurun_out/r03_pytest_decode.log ;;
    ppo_ab)      # the PPO iteration with and without the fused decode epilogues, same box
      AA_DECODE_EPI=0 timeout 600 python tools/bench_ppo.py > gpurun_out/r03_bench_ppo_epi0.json 2> gpurun_out/r03_bench_ppo_epi0.err; python -c "import json; d=json.load(open('gpurun_out/r03_bench_ppo_epi0.json')); print('EPI=0', d['iteration_ms'], d['decode_ms_per_position'], d['split_ms'])"
      timeout 600 python tools/bench_ppo.py > gpurun_out/r03_bench_ppo_epi1.json 2> gpurun_out/r03_bench_ppo_epi1.err; python -c "import json; d=json.load(open('gpurun_out/r03_bench_ppo_epi1.json')); print('EPI=1', d['iteration_ms'], d['decode_ms_per_position'], d['split_ms'])"
      timeout 600 python tools/bench_decode.py > gpurun_out/r03_bench_decode_7b.json 2> gpurun_out/r03_bench_decode.err; tail -c 1200 gpurun_out/r03_bench_decode_7b.json ;;
    full_depth)
      timeout 900 python -m pytest tests/test_secondary_geometry_gpu.py -m gpu -q -p no:cacheprovider -k full_depth > gpurun_out/r03_pytest_full_depth.log 2>&1; tail -12 gpurun_out/r03_pytest_full_depth.log; cat gpurun_out/parity_full_depth_llava7b_bf16_vs_twin.txt ;;
    attn_lab)    # same-box A/B of attention builds (tools/build_attn_variant.sh): libaa_hip.so = production, _prio1 / _prio2 = s_setprio variants
      AA_ATTN_LIBS=${AA_ATTN_LIBS:-libaa_hip.so,libaa_hip_prio1.so,libaa_hip_prio2.so,libaa_hip.so} AA_LAB_OUT=r03_attention_lab.json timeout 600 python tools/attn_lab.py > gpurun_out/r03_attention_lab.txt 2>&1; cat gpurun_out/r03_attention_lab.txt ;;
    decode_lab)  # PPO iteration per library in AA_DECODE_LIBS (strip-kernel lab builds), same box
      for lib in ${AA_DECODE_LIBS:-libaa_hip.so}; do
        AA_HIP_LIB=$R/align_anything_amd/$lib timeout 600 python tools/bench_ppo.py --iters 2 > gpurun_out/r03_ppo_$lib.json 2> gpurun_out/r03_ppo_$lib.err
        python -c "import json,sys; d=json.load(open('gpurun_out/r03_ppo_$lib.json')); print('$lib', 'decode ms/pos', round(d['decode_ms_per_position'],4), 'iteration', round(d['iteration_ms'],1))" || tail -3 gpurun_out/r03_ppo_$lib.err
      done ;;
    bench_ab)    # the headline step per library in AA_BENCH_LIBS (same box, alternating, twice)
      for rep in 1 2; do for lib in ${AA_BENCH_LIBS:-libaa_hip.so}; do
        AA_HIP_LIB=$R/align_anything_amd/$lib timeout 600 python bench.py --steps 6 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > gpurun_out/r03_bench_ab_$lib.json 2> gpurun_out/r03_bench_ab_$lib.err
        python -c "import json; d=json.load(open('gpurun_out/r03_bench_ab_$lib.json')); k=[x for x in d['roofline']['by_kind_top12'] if x['tflop'] in (2.9549, 1.4775)]; print('$lib rep $rep', round(d['ms_per_step'],2), round(d['value'],4), d.get('glu_bwd_plan',[{}])[0], [(x['algorithmic_MB'], x['avg_ms']) for x in k])" || tail -3 gpurun_out/r03_bench_ab_$lib.err
      done; done ;;
    unit_tests)  # kernel-level suites after an arithmetic change
      timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_elementwise_gpu.py tests/test_twin_gpu.py tests/test_decode_gpu.py tests/test_model_gpu.py tests/test_bench_geometry_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r03_pytest_unit.log 2>&1; tail -6 gpurun_out/r03_pytest_unit.log ;;
    ppo_prof)    # kernel trace of one PPO iteration (decode kernel split at the Qwen2-VL-7B geometry)
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r03_ppo_prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_ppo_prof -o p -- python $R/tools/bench_ppo.py --iters 1 --new-tokens 256 > $R/gpurun_out/r03_bench_ppo_under_rocprof.json 2> $R/gpurun_out/r03_ppo_prof.err )
      f=$(find gpurun_out/r03_ppo_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r03_ppo_kernel_stats.csv && head -22 "$f" | cut -c1-260
      find gpurun_out/r03_ppo_prof -name "*kernel_trace.csv" -delete ;;
    ppo4)
      timeout 900 python tools/bench_ppo.py --prompts 4 > gpurun_out/r03_bench_ppo_4prompts.json 2> gpurun_out/r03_bench_ppo4.err; tail -c 1500 gpurun_out/r03_bench_ppo_4prompts.json; tail -5 gpurun_out/r03_bench_ppo4.err ;;
    ppo)
      timeout 1500 python tools/bench_ppo.py > gpurun_out/r03_bench_ppo.json 2> gpurun_out/r03_bench_ppo.err; tail -c 2500 gpurun_out/r03_bench_ppo.json; tail -5 gpurun_out/r03_bench_ppo.err ;;
    rope_ab)     # attention backward with the rotary backward in its epilogues (aa_attn_bwd_rope) vs the separate launch: bit-identity tests, then the headline step both ways, same box
      timeout 600 python -m pytest tests/test_attention_gpu.py tests/test_model_gpu.py tests/test_qwen3moe_gpu.py tests/test_twin_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r03_pytest_rope.log 2>&1; tail -6 gpurun_out/r03_pytest_rope.log
      for rep in 1 2; do for v in 0 1; do
        AA_ATTN_ROPE=$v timeout 600 python bench.py --steps 6 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > gpurun_out/r03_bench_rope$v.json 2> gpurun_out/r03_bench_rope$v.err
        python -c "import json; d=json.load(open('gpurun_out/r03_bench_rope$v.json')); print('AA_ATTN_ROPE=$v rep $rep', round(d['ms_per_step'],2), round(d['value'],4))" || tail -3 gpurun_out/r03_bench_rope$v.err
      done; done ;;
    ep_tests)    # MoE kernels with pos = -1 rows, the capacity-padded expert-parallel exchange (no host sync; 2 ranks == exact exchange bit for bit)
      timeout 900 python -m pytest tests/test_qwen3moe_gpu.py tests/test_ep_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r03_pytest_ep.log 2>&1; tail -30 gpurun_out/r03_pytest_ep.log; cat gpurun_out/parity/parity_expert_parallel.txt 2>/dev/null ;;
    moe_prof)    # kernel split of the Qwen3-30B-A3B-geometry DPO step (12 layers)
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r03_moe_prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_moe_prof -o p -- python $R/tools/bench_qwen3moe.py --steps 3 --warmup 1 > $R/gpurun_out/r03_bench_qwen3moe_under_rocprof.json 2> $R/gpurun_out/r03_moe_prof.err )
      f=$(find gpurun_out/r03_moe_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r03_qwen3moe_kernel_stats.csv && head -40 "$f" | cut -c1-200
      find gpurun_out/r03_moe_prof -name "*kernel_trace.csv" -delete; tail -3 gpurun_out/r03_moe_prof.err ;;
    moe_bench)   # the 12-layer Qwen3-30B-A3B-geometry DPO step
      timeout 600 python tools/bench_qwen3moe.py --steps 4 --warmup 2 > gpurun_out/r03_bench_qwen3moe.json 2> gpurun_out/r03_bench_qwen3moe.err; cat gpurun_out/r03_bench_qwen3moe.json; tail -3 gpurun_out/r03_bench_qwen3moe.err ;;
    *) echo "unknown stage $stage" ;;
  esac
done
echo "=== done $(date +%T)"
