#!/bin/bash
# Round-5 GPU stages.  Usage on the GPU box: bash tools/gpu_r5.sh <stage> [<stage> ...]; every stage writes under gpurun_out/ (merged back by gpurun).
# The stages of experiments that were measured and removed again (persistent decode, sampler unroll, tower prefetch, MoE dW on the gemm4 tile, the AdamW lab A/B)
# went with their code: their numbers are in profiles/r05_*_negative.txt / r05_adam_window.txt, their commands in git history.
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"; mkdir -p gpurun_out
for stage in "$@"; do
  echo "=== stage $stage  $(date +%T)"
  case $stage in
    attn_pmc_instep)   # clock-vs-fabric for the in-step attention kernels: GRBM_GUI_ACTIVE + SQ_BUSY_CYCLES (effective clock = cycles / trace duration) and FETCH_SIZE, three separate --pmc passes over the bench step itself
      ( cd /tmp && export TMPDIR=/tmp
        for set in "GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES" "FETCH_SIZE"; do
          rm -rf $R/gpurun_out/r05_pmc_$set
          timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/r05_pmc_$set -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-events --traffic committed --no-per-batch > $R/gpurun_out/r05_pmc_$set.log 2>&1
        done )
      python3 tools/pmc_instep_summary.py gpurun_out > gpurun_out/r05_attn_instep_pmc.txt; cat gpurun_out/r05_attn_instep_pmc.txt | cut -c1-220
      find gpurun_out/r05_pmc_* -name "*kernel_trace.csv" -delete; find gpurun_out/r05_pmc_* -name "*counter_collection.csv" -size +8M -delete ;;
    dropin)          # the end-to-end drop-in test and the derived bf16 envelope at width
      timeout 900 python -m pytest tests/test_dropin_gpu.py tests/test_secondary_geometry_gpu.py -q -x -m gpu -p no:cacheprovider -k "drop_in or llava7b_width" > gpurun_out/r05_dropin.log 2>&1; echo "rc=$?"; tail -30 gpurun_out/r05_dropin.log | cut -c1-400 ;;
    norm)            # RMSNorm forward / backward at the step's shapes + their numerics tests
      timeout 300 python tools/bench_norm.py r05_bench_norm.json 2>&1 | cut -c1-200
      timeout 600 python -m pytest tests/test_elementwise_gpu.py tests/test_twin_gpu.py -q -x -m gpu -p no:cacheprovider -k "norm" > gpurun_out/r05_norm_tests.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r05_norm_tests.log | cut -c1-300 ;;
    bench_quick)     # headline step without the PMC passes / CPU leg / batch sweep; power + clock sidecar on
      timeout 600 python bench.py --steps 6 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > gpurun_out/r05_bench_quick.json 2> gpurun_out/r05_bench_quick.err; python -c "import json; d=json.load(open('gpurun_out/r05_bench_quick.json')); r=d['roofline']; print('ms/step', d['ms_per_step'], 'pairs/s', d['value'], 'gemm frac', r['frac'], {k: r.get(k) for k in ('power_source','power_w_mean','sclk_mhz_mean','peak_at_sclk','frac_of_peak_at_sclk','j_per_tflop_step','power_samples','power_source_errors')})" || tail -5 gpurun_out/r05_bench_quick.err
      python tools/power_sampler.py --out gpurun_out/r05_power_probe.jsonl --seconds 1; head -c 1500 gpurun_out/r05_power_probe.jsonl ;;
    tests)
      timeout 1700 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/r05_pytest.log 2>&1; tail -15 gpurun_out/r05_pytest.log ;;
    dp_shadow)       # one-GPU model of a resident collective beside backward (CU-masked stream, traffic kernel)
      timeout 900 python tools/dp_shadow.py --steps 4 > gpurun_out/r05_dp_shadow.log 2> gpurun_out/r05_dp_shadow.err; echo "rc=$?"; cut -c1-200 gpurun_out/r05_dp_shadow.log; tail -3 gpurun_out/r05_dp_shadow.err | cut -c1-300 ;;
    attn_pair)       # round 5: paired query blocks + XCD-local order of the head_dim-128 forward (AA_ATTN128 bit 2) vs kv heads fastest: numerics, standalone, in-step, fetch
      timeout 400 python tools/attn128_check.py --base 3 --impl 7 --out r05_attn_pair_check.json > gpurun_out/r05_attn_pair_check.txt 2>&1; python3 - <<'PY'
import json
for c in json.load(open('gpurun_out/r05_attn_pair_check.json')):
    print(c['case'], 'ok' if c['ok'] else 'MISMATCH', {k: v for k, v in c.items() if 'identical' in k or k.endswith('_us')})
PY
      tail -1 gpurun_out/r05_attn_pair_check.txt
      for rep in 1 2; do for v in 3 7; do
        AA_ATTN128=$v timeout 600 python bench.py --steps 6 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > gpurun_out/r05_bench_attn$v.json 2> gpurun_out/r05_bench_attn$v.err
        python -c "import json; d=json.load(open('gpurun_out/r05_bench_attn$v.json')); r=d['roofline']; print('AA_ATTN128=$v rep $rep', round(d['ms_per_step'],2), round(d['value'],4), 'W', round(r.get('power_w_mean') or 0), 'MHz', round(r.get('sclk_mhz_mean') or 0))" || tail -3 gpurun_out/r05_bench_attn$v.err
      done; done
      ( cd /tmp && export TMPDIR=/tmp
        for v in 3 7; do
          rm -rf $R/gpurun_out/r05_pmc_attn$v
          AA_ATTN128=$v timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r05_pmc_attn$v -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-events --traffic committed --no-per-batch --no-power > $R/gpurun_out/r05_pmc_attn$v.log 2>&1
          python3 - $R/gpurun_out/r05_pmc_attn$v <<'PY'
import csv, glob, sys, collections
dur = {}
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'attn' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            acc[r['Kernel_Name'].split('(')[0][-40:]].append((float(r['Counter_Value']), dur.get(r['Dispatch_Id'], 0.0)))
for k, v in acc.items():
    fs = sum(x for x, _ in v) / len(v); us = sum(y for _, y in v) / len(v)
    print(sys.argv[1][-5:], k, 'n', len(v), 'avg us', round(us, 1), 'fetch GB (2 x KiB)', round(fs * 2048 / 1e9, 3))
PY
          find $R/gpurun_out/r05_pmc_attn$v -name "*.csv" -size +4M -delete
        done ) ;;
    decode)          # the decode / sampling tests and the PPO iteration
      timeout 900 python -m pytest tests/test_decode_gpu.py -q -x -m gpu -p no:cacheprovider > gpurun_out/r05_decode_tests.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r05_decode_tests.log | cut -c1-300
      timeout 300 python tools/bench_ppo.py --iters 2 > gpurun_out/r05_bench_ppo.json 2> gpurun_out/r05_bench_ppo.err
      python -c "import json; d=json.load(open('gpurun_out/r05_bench_ppo.json')); print('decode', round(d['decode_ms_per_position'],4), 'ms/pos', round(d['iteration_ms'],1), 'ms/iter', d['split_ms'])" || tail -3 gpurun_out/r05_bench_ppo.err ;;
    moe)             # configs[4] at 2 (the number quoted since round 2) and 4 pairs per step
      for b in 2 4; do
        timeout 400 python tools/bench_qwen3moe.py --pairs $b --steps 4 --warmup 2 > gpurun_out/r05_bench_qwen3moe_b$b.json 2> gpurun_out/r05_bench_qwen3moe_b$b.err; cut -c1-500 gpurun_out/r05_bench_qwen3moe_b$b.json; tail -2 gpurun_out/r05_bench_qwen3moe_b$b.err
      done ;;
    decode_stats)    # per-kernel averages of the decode window (sampler, strip GEMVs) under rocprofv3
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r05_ppo_prof && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_ppo_prof -o p -- python $R/tools/bench_ppo.py --iters 1 --new-tokens 128 > $R/gpurun_out/r05_bench_ppo_under_rocprof.json 2> $R/gpurun_out/r05_ppo_prof.err )
      f=$(find gpurun_out/r05_ppo_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r05_ppo_kernel_stats.csv && grep -i "sample\|skinny\|attn_decode\|argmax" "$f" | cut -c1-60,200-330
      find gpurun_out/r05_ppo_prof -name "*kernel_trace.csv" -delete ;;
    llama31)         # the reference's default text backbone at width vs the reference trainer's fixture
      timeout 600 python -m pytest tests/test_llama3_gpu.py -q -x -m gpu -p no:cacheprovider -k "width" > gpurun_out/r05_llama31_width.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/r05_llama31_width.log | cut -c1-400; cat gpurun_out/parity_llama31_width_vs_reference.txt | cut -c1-300 ;;
    bench_b1)        # the headline step at the reference yaml's micro-batch (1 pair): full roofline split at M = 4096
      timeout 600 python bench.py --pairs-per-gpu 1 --steps 8 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > gpurun_out/r05_bench_b1.json 2> gpurun_out/r05_bench_b1.err; python -c "
import json; d=json.load(open('gpurun_out/r05_bench_b1.json')); r=d['roofline']; print('B1 ms/step', d['ms_per_step'], 'pairs/s', d['value'], 'gemm frac', r['frac'], 'share', r['gemm_share_of_step_time'], 'W', r.get('power_w_mean'), 'MHz', r.get('sclk_mhz_mean'))
for k in r['by_kind_top12'][:8]: print(k)" || tail -5 gpurun_out/r05_bench_b1.err ;;
    bench)           # the driver's command (defaults: in-run PMC traffic, CPU leg, B1 / B2 datapoints, power + clock sidecar)
      timeout 1500 python bench.py --steps 8 --warmup 2 > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; tail -c 1800 gpurun_out/r05_bench.json; tail -5 gpurun_out/r05_bench.err ;;
    prof)            # the same step under rocprofv3 --kernel-trace --stats: the per-kernel averages the roofline line is checked against
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r05_prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof -o p -- python $R/bench.py --steps 3 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > $R/gpurun_out/r05_bench_under_rocprof.json 2> $R/gpurun_out/r05_prof.err )
      f=$(find gpurun_out/r05_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r05_dpo7b_kernel_stats.csv && head -22 "$f" | cut -c1-200
      find gpurun_out/r05_prof -name "*kernel_trace.csv" -delete ;;
    secondary)       # the secondary configs at 4 pairs per step + PPO iteration (final code)
      timeout 400 python tools/bench_qwen2vl.py --pairs 4 --steps 3 --warmup 1 > gpurun_out/r05_bench_qwen2vl_b4.json 2> gpurun_out/r05_bench_qwen2vl_b4.err; cut -c1-400 gpurun_out/r05_bench_qwen2vl_b4.json; tail -2 gpurun_out/r05_bench_qwen2vl_b4.err
      timeout 400 python tools/bench_qwen2audio.py --pairs 4 --steps 3 --warmup 1 > gpurun_out/r05_bench_qwen2audio_b4.json 2> gpurun_out/r05_bench_qwen2audio_b4.err; cut -c1-400 gpurun_out/r05_bench_qwen2audio_b4.json; tail -2 gpurun_out/r05_bench_qwen2audio_b4.err ;;
    final)           # round end: smoke(), the N = 2 code path of bench.py as a FUNCTIONAL run on one device (gloo; never a performance number), the full suite
      timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
      AA_BENCH_ONE_DEVICE=1 AA_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 2 --warmup 1 --layers 4 --no-cpu-baseline --no-gemm-events > gpurun_out/r05_dp2_functional_onebox.json 2> gpurun_out/r05_dp2_functional_onebox.err; python -c "import json; d=json.loads(open('gpurun_out/r05_dp2_functional_onebox.json').read().strip().split(chr(10))[-1]); m=d['multi_gpu']; print('dp2 functional:', d['config']['workload'][-60:], 'replicas identical', m['replicas_bit_identical_after_steps'], 'reduce', m['reduce_mode'], (m.get('reduce_autotune') or {}).get('forms_agree'))" || tail -5 gpurun_out/r05_dp2_functional_onebox.err
      timeout 1700 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/r05_pytest.log 2>&1; tail -8 gpurun_out/r05_pytest.log ;;
    qwen2vl_width)   # BASELINE configs[2]'s backbone at width vs the reference trainer's fixture
      timeout 900 python -m pytest tests/test_qwen2vl_gpu.py -q -x -m gpu -p no:cacheprovider -k "width_pair" > gpurun_out/r05_qwen2vl_width.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/r05_qwen2vl_width.log | cut -c1-400; cat gpurun_out/parity_qwen2vl_width_vs_reference.txt | cut -c1-400 ;;
    widths)          # the configs[3] / [4] backbones at width vs the reference trainer's fixtures
      timeout 1200 python -m pytest tests/test_qwen2audio_gpu.py tests/test_qwen3moe_gpu.py -q -m gpu -p no:cacheprovider -k "width_pair" > gpurun_out/r05_widths.log 2>&1; echo "rc=$?"; tail -14 gpurun_out/r05_widths.log | cut -c1-400
      cat gpurun_out/parity_qwen2audio_width_vs_reference.txt gpurun_out/parity_qwen3moe_width_vs_reference.txt | cut -c1-400 ;;
    energy_ab)       # the two open energy questions under the power / clock fields: gemm4's tile-group height (AA_GEMM_GM 3 / 4 / 8; default =
                     # the shape heuristic) and the attention forward's defer-max threshold (lab library libaa_hip_thr0.so: AT_THR = 0 = rescale on every new maximum)
      for rep in 1 2; do
        for gm in 0 3 4 8; do
          AA_GEMM_GM=$gm timeout 600 python bench.py --steps 6 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > gpurun_out/r05_bench_gm$gm.json 2> gpurun_out/r05_bench_gm$gm.err
          python -c "import json; d=json.load(open('gpurun_out/r05_bench_gm$gm.json')); r=d['roofline']; print('AA_GEMM_GM=$gm rep $rep', round(d['ms_per_step'],2), 'ms  gemm4', round(r['achieved']), 'TF/s  W', round(r['power_w_mean']), 'MHz', round(r['sclk_mhz_mean']), 'J/TFLOP', round(r['j_per_tflop_step'],4), 'frac@sclk', round(r['frac_of_peak_at_sclk'],4))" || tail -3 gpurun_out/r05_bench_gm$gm.err
        done
        for lib in libaa_hip.so libaa_hip_thr0.so; do
          AA_HIP_LIB=$R/align_anything_amd/$lib timeout 600 python bench.py --steps 6 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > gpurun_out/r05_bench_$lib.json 2> gpurun_out/r05_bench_$lib.err
          python -c "import json; d=json.load(open('gpurun_out/r05_bench_$lib.json')); r=d['roofline']; print('$lib rep $rep', round(d['ms_per_step'],2), 'ms  W', round(r['power_w_mean']), 'MHz', round(r['sclk_mhz_mean']), 'J/TFLOP', round(r['j_per_tflop_step'],4))" || tail -3 gpurun_out/r05_bench_$lib.err
        done
      done ;;
    sampler)         # round 5: the sampler's exponentials on v_exp_f32: tests, per-kernel time, PPO iteration
      timeout 600 python -m pytest tests/test_decode_gpu.py tests/test_grpo_gpu.py tests/test_ppo_gpu.py tests/test_dropin_gpu.py -q -x -m gpu -p no:cacheprovider -k "not end_to_end" > gpurun_out/r05_sampler_tests.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r05_sampler_tests.log | cut -c1-300
      python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from align_anything_amd import ops
from tools.bench_kernels import timeit
dev = torch.device('cuda:0')
for V in (152064, 32064, 128256):
    lg = (torch.randn(1, V, device=dev) * 3).to(torch.bfloat16)
    u = torch.rand(1, device=dev)
    for top_p, top_k in ((1.0, 0), (0.9, 50)):
        ms = timeit(lambda: ops.sample_top_p(lg, 1.0, top_p, u, None, 1.0, top_k=top_k), iters=200, warm=20)
        print('V', V, 'top_p', top_p, 'top_k', top_k, 'us', round(ms * 1e3, 1), flush=True)
PY
      for v in 1 2; do
        timeout 300 python tools/bench_ppo.py --iters 2 > gpurun_out/r05_bench_ppo_sampler$v.json 2> gpurun_out/r05_bench_ppo_sampler$v.err
        python -c "import json; d=json.load(open('gpurun_out/r05_bench_ppo_sampler$v.json')); print('run $v', round(d['decode_ms_per_position'],4), 'ms/pos', round(d['iteration_ms'],1), 'ms/iter')" || tail -3 gpurun_out/r05_bench_ppo_sampler$v.err
      done ;;
    *) echo "unknown stage $stage" ;;
  esac
done
echo "=== done $(date +%T)"
