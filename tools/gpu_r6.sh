#!/bin/bash
# Round-6 GPU stages.  Usage on the GPU box: bash tools/gpu_r6.sh <stage> [<stage> ...]; every stage writes under gpurun_out/ (merged back by gpurun).
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"; mkdir -p gpurun_out
for stage in "$@"; do
  echo "=== stage $stage  $(date +%T)"
  case $stage in
    full_depth)      # the headline configuration at L = 32 against the reference trainer's fixture (fp32 twin + derived bf16 envelope)
      timeout 2400 python -m pytest tests/test_secondary_geometry_gpu.py -q -m gpu -p no:cacheprovider -k "full_depth_pair" > gpurun_out/r06_full_depth.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/r06_full_depth.log | cut -c1-400
      cat gpurun_out/parity_llava7b_full_depth_vs_reference.txt gpurun_out/parity_llava7b_full_depth_packed_vs_reference.txt | cut -c1-500 ;;
    atomic)          # lab: fp32 global atomic throughput in the dQ pattern of a single-pass attention backward
      hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lab/ubench/atomic_f32.hip -o /tmp/atomic_f32 && timeout 120 /tmp/atomic_f32 | tee gpurun_out/r06_atomic_f32.txt ;;
    tests)
      timeout 2400 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --deselect tests/test_secondary_geometry_gpu.py::test_llava7b_full_depth_pair_vs_the_reference_trainer --deselect tests/test_secondary_geometry_gpu.py::test_llava7b_full_depth_pair_with_shared_prompt_packing_vs_the_reference_trainer > gpurun_out/r06_pytest.log 2>&1; tail -15 gpurun_out/r06_pytest.log | cut -c1-300 ;;
    bench)           # the driver's command, full line
      timeout 900 python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "rc=$?"
      python - <<'PY'
import json
d = json.load(open('gpurun_out/r06_bench.json')); r = d['roofline']
print('ms/step', round(d['ms_per_step'], 2), 'pairs/s', round(d['value'], 4), 'gemm frac', round(r['frac'], 4), 'W', r.get('power_w_mean'), 'MHz', r.get('sclk_mhz_mean'))
for k in r.get('hbm_kernels', []):
    print('  ', k['kernel'][:44], 'n', k['sampled_launches'], 'MB', round(k['algorithmic_bytes'] / 1e6, 1), 'ms', round(k['avg_ms'], 4), 'frac', round(k['frac_of_8TBs'], 3), 'ms/step', round(k['ms_per_step'], 2))
print('  attention', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.get('attention', {}).items() if k != 'note'})
print('  per_batch', {k: round(v['value'], 3) for k, v in d.get('per_batch', {}).items() if isinstance(v, dict)})
PY
      ;;
    bench_quick)     # headline step without the PMC passes / CPU leg / batch sweep
      timeout 600 python bench.py --steps 8 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > gpurun_out/r06_bench_quick.json 2> gpurun_out/r06_bench_quick.err
      python -c "import json; d=json.load(open('gpurun_out/r06_bench_quick.json')); r=d['roofline']; print('ms/step', d['ms_per_step'], 'pairs/s', d['value'], 'gemm frac', r['frac'], 'W', r.get('power_w_mean'), 'MHz', r.get('sclk_mhz_mean')); print(r.get('attention'))" || tail -5 gpurun_out/r06_bench_quick.err ;;
    rocprof)         # rocprofv3 --kernel-trace --stats of the bench step: per-kernel averages for profiles/r06_dpo7b_kernel_stats.csv
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r06_prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_prof -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-events --traffic committed --no-per-batch --no-power > $R/gpurun_out/r06_prof.log 2>&1 )
      f=$(find gpurun_out/r06_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r06_dpo7b_kernel_stats.csv; head -25 gpurun_out/r06_dpo7b_kernel_stats.csv | cut -c1-200
      find gpurun_out/r06_prof -name "*kernel_trace.csv" -delete ;;
    glue)            # which python lines launch torch copy / fill kernels inside the step
      timeout 600 python tools/lab/glue_prof.py 8 2>&1 | tail -90 | cut -c1-200 ;;
    decode)          # decode of the FINAL code at the batches the reference rolls out (PPO: 1; GRPO: B x num_generations = 10, grpo.py:216-226, grpo.yaml:70) and 4 / 16
      AA_BENCH_DECODE_CASES="1,512,64;4,512,64;10,512,64;16,512,64" AA_BENCH_DECODE_OUT=r06_bench_decode.json timeout 900 python tools/bench_decode.py 2>&1 | grep -E "^\{" | cut -c1-330 ;;
    ppo)             # BASELINE configs[2] on one GPU: Qwen2-VL-7B actor + reference + reward + critic, one PPO iteration
      timeout 900 python tools/bench_ppo.py --iters 2 > gpurun_out/r06_bench_ppo.log 2>&1; tail -4 gpurun_out/r06_bench_ppo.log | cut -c1-600 ;;
    moe_tie)
      timeout 300 python -m pytest tests/test_qwen3moe_gpu.py -q -x -m gpu -p no:cacheprovider -k "tie_rule or kernels_vs_torch" 2>&1 | tail -4 ;;
    scc_ab)          # the SCC clobber on the LDS-DMA asm statements (default since round 6) against the old build (libaa_hip_noscc.so): numerics of the touched kernels, then the headline step, alternating
      timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_attention_gpu.py tests/test_bench_geometry_gpu.py -q -x -m gpu -p no:cacheprovider > gpurun_out/r06_scc_tests.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r06_scc_tests.log | cut -c1-200
      for v in libaa_hip_noscc.so libaa_hip.so libaa_hip_noscc.so libaa_hip.so; do
        AA_HIP_LIB=$R/align_anything_amd/$v timeout 600 python bench.py --steps 8 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > gpurun_out/r06_bench_scc.json 2> gpurun_out/r06_bench_scc.err
        python -c "import json; d=json.load(open('gpurun_out/r06_bench_scc.json')); r=d['roofline']; print('$v', round(d['ms_per_step'],2), 'ms', round(d['value'],4), 'pairs/s  gemm4', round(r['achieved'],1), 'TF/s  W', round(r.get('power_w_mean') or 0), 'MHz', round(r.get('sclk_mhz_mean') or 0), 'losses', d['config'].get('losses_timed_steps', [])[-2:])" || tail -3 gpurun_out/r06_bench_scc.err
      done ;;
    pack)            # shared-prompt packing (train_cfgs.share_prompt_prefix): packed vs unpacked, packed vs the reference trainer at full depth, then the bench line (its `shared_prompt` section)
      timeout 600 python -m pytest tests/test_pack_gpu.py -q -x -m gpu -p no:cacheprovider > gpurun_out/r06_pack_tests.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/r06_pack_tests.log | cut -c1-300; cat gpurun_out/parity_pack_*.txt
      if [ -z "$AA_PACK_SKIP_FULL" ]; then timeout 1500 python -m pytest tests/test_secondary_geometry_gpu.py -q -x -m gpu -p no:cacheprovider -k "shared_prompt_packing" > gpurun_out/r06_pack_full_depth.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/r06_pack_full_depth.log | cut -c1-300; cut -c1-400 gpurun_out/parity_llava7b_full_depth_packed_vs_reference.txt; fi
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r06_prof_pack && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_prof_pack -o p -- python $R/bench.py --share-prompt --steps 4 --warmup 2 --no-cpu-baseline --no-gemm-events --traffic committed --no-per-batch --no-power > $R/gpurun_out/r06_prof_pack.log 2>&1 )
      f=$(find gpurun_out/r06_prof_pack -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r06_dpo7b_packed_kernel_stats.csv; head -22 gpurun_out/r06_dpo7b_packed_kernel_stats.csv | cut -c1-170; tail -2 gpurun_out/r06_prof_pack.log | cut -c1-300
      find gpurun_out/r06_prof_pack -name "*kernel_trace.csv" -delete
      timeout 900 python bench.py --steps 8 --warmup 2 --traffic committed --no-cpu-baseline > gpurun_out/r06_bench_pack.json 2> gpurun_out/r06_bench_pack.err; echo "rc=$?"
      python -c "import json; d=json.load(open('gpurun_out/r06_bench_pack.json')); print('headline', round(d['ms_per_step'],2), 'ms', round(d['value'],4), 'pairs/s'); print('shared_prompt', {k: (round(v,4) if isinstance(v,float) else v) for k,v in d['shared_prompt'].items() if k != 'note'}); print(d['config']['losses_timed_steps'][-2:])" || tail -5 gpurun_out/r06_bench_pack.err ;;
    secondary)       # the other 7B backbones at 4 pairs per step + the MoE step, on the final code (tile-model change, SCC clobber)
      timeout 400 python tools/bench_qwen2vl.py --pairs 4 --steps 3 --warmup 1 > gpurun_out/r06_bench_qwen2vl_b4.json 2> gpurun_out/r06_bench_qwen2vl_b4.err; cut -c1-420 gpurun_out/r06_bench_qwen2vl_b4.json; tail -2 gpurun_out/r06_bench_qwen2vl_b4.err | cut -c1-200
      timeout 400 python tools/bench_qwen2audio.py --pairs 4 --steps 3 --warmup 1 > gpurun_out/r06_bench_qwen2audio_b4.json 2> gpurun_out/r06_bench_qwen2audio_b4.err; cut -c1-420 gpurun_out/r06_bench_qwen2audio_b4.json; tail -2 gpurun_out/r06_bench_qwen2audio_b4.err | cut -c1-200
      for b in 2 4; do timeout 400 python tools/bench_qwen3moe.py --pairs $b --steps 4 --warmup 2 > gpurun_out/r06_bench_qwen3moe_b$b.json 2> gpurun_out/r06_bench_qwen3moe_b$b.err; cut -c1-420 gpurun_out/r06_bench_qwen3moe_b$b.json; tail -2 gpurun_out/r06_bench_qwen3moe_b$b.err | cut -c1-200; done ;;
    pack_audio)      # shared-prompt packing on the Qwen2-Audio DPO path (BASELINE configs[3] backbone): packed against unpacked (fp32 twin), then the step, both ways
      timeout 600 python -m pytest tests/test_pack_gpu.py tests/test_qwen2audio_gpu.py -q -x -m gpu -p no:cacheprovider > gpurun_out/r06_pack_audio_tests.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/r06_pack_audio_tests.log | cut -c1-300; cat gpurun_out/parity_pack_qwen2audio.txt
      for f in "" "--share-prompt"; do timeout 400 python tools/bench_qwen2audio.py --pairs 4 --steps 3 --warmup 1 $f > gpurun_out/r06_bench_qwen2audio_b4$f.json 2> gpurun_out/r06_bench_qwen2audio_b4$f.err; cut -c1-520 gpurun_out/r06_bench_qwen2audio_b4$f.json; tail -2 gpurun_out/r06_bench_qwen2audio_b4$f.err | cut -c1-200; done ;;
    tail)            # dead-row elimination of the last decoder layer: consumed numbers unchanged (tests), then the headline step with and without it, alternating on this box
      timeout 900 python -m pytest tests/test_tail_gpu.py tests/test_pack_gpu.py tests/test_model_gpu.py tests/test_twin_gpu.py -q -x -m gpu -p no:cacheprovider > gpurun_out/r06_tail_tests.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r06_tail_tests.log | cut -c1-300; cat gpurun_out/parity_tail_prune_*.txt
      for v in 0 1 0 1; do
        AA_TAIL_PRUNE=$v timeout 600 python bench.py --steps 8 --warmup 2 --traffic committed --no-cpu-baseline --no-per-batch > gpurun_out/r06_bench_tail$v.json 2> gpurun_out/r06_bench_tail$v.err
        python -c "import json; d=json.load(open('gpurun_out/r06_bench_tail$v.json')); r=d['roofline']; sp=d.get('shared_prompt',{}); print('AA_TAIL_PRUNE=$v', round(d['ms_per_step'],2), 'ms', round(d['value'],4), 'pairs/s  W', round(r.get('power_w_mean') or 0), 'MHz', round(r.get('sclk_mhz_mean') or 0), 'executed TF/pair', round(d['step_mfma']['executed_tflop_per_pair'],2), 'packed', round(sp.get('ms_per_step',0),2), 'ms losses', d['config'].get('losses_timed_steps', [])[-2:])" || tail -3 gpurun_out/r06_bench_tail$v.err
      done ;;
    pack_vl)         # shared-prompt packing on the Qwen2-VL DPO path (the configs[2] backbone; multimodal RoPE): packed against unpacked (fp32 twin), then the step, both ways
      timeout 600 python -m pytest tests/test_pack_gpu.py tests/test_qwen2vl_gpu.py -q -x -m gpu -p no:cacheprovider > gpurun_out/r06_pack_vl_tests.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/r06_pack_vl_tests.log | cut -c1-300; cat gpurun_out/parity_pack_qwen2vl*.txt
      for f in "" "--share-prompt"; do timeout 400 python tools/bench_qwen2vl.py --pairs 4 --steps 3 --warmup 1 $f > gpurun_out/r06_bench_qwen2vl_b4$f.json 2> gpurun_out/r06_bench_qwen2vl_b4$f.err; cut -c1-520 gpurun_out/r06_bench_qwen2vl_b4$f.json; tail -2 gpurun_out/r06_bench_qwen2vl_b4$f.err | cut -c1-200; done ;;
    glue_moe)        # torch ops inside one Qwen3-MoE DPO step, by python site
      timeout 600 python tools/lab/glue_prof.py moe 2>&1 | tail -70 | cut -c1-220 ;;
    rocprof_moe)     # rocprofv3 --kernel-trace --stats of the Qwen3-MoE step (tools/bench_qwen3moe.py, 2 pairs)
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r06_prof_moe && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_prof_moe -o p -- python $R/tools/bench_qwen3moe.py --pairs 2 --steps 4 --warmup 2 > $R/gpurun_out/r06_prof_moe.log 2>&1 )
      f=$(find gpurun_out/r06_prof_moe -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r06_qwen3moe_kernel_stats.csv; head -40 gpurun_out/r06_qwen3moe_kernel_stats.csv | cut -c1-190; tail -1 gpurun_out/r06_prof_moe.log | cut -c1-300
      find gpurun_out/r06_prof_moe -name "*kernel_trace.csv" -delete ;;
    pack_moe)        # shared-prompt packing on the Qwen3-MoE DPO path (single-rank experts): packed against unpacked, then the step, both ways
      timeout 600 python -m pytest tests/test_pack_gpu.py tests/test_qwen3moe_gpu.py -q -x -m gpu -p no:cacheprovider > gpurun_out/r06_pack_moe_tests.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/r06_pack_moe_tests.log | cut -c1-300; cat gpurun_out/parity_pack_qwen3moe*.txt
      for f in "" "--share-prompt"; do timeout 400 python tools/bench_qwen3moe.py --pairs 2 --steps 4 --warmup 2 $f > gpurun_out/r06_bench_qwen3moe_b2$f.json 2> gpurun_out/r06_bench_qwen3moe_b2$f.err; cut -c1-520 gpurun_out/r06_bench_qwen3moe_b2$f.json; tail -2 gpurun_out/r06_bench_qwen3moe_b2$f.err | cut -c1-200; done
      timeout 400 python tools/bench_qwen3moe.py --pairs 4 --steps 4 --warmup 2 --share-prompt > gpurun_out/r06_bench_qwen3moe_b4--share-prompt.json 2> gpurun_out/r06_bench_qwen3moe_b4--share-prompt.err; cut -c1-520 gpurun_out/r06_bench_qwen3moe_b4--share-prompt.json ;;
    rocprof_ppo)     # rocprofv3 --kernel-trace --stats of the PPO iteration (tools/bench_ppo.py)
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r06_prof_ppo && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_prof_ppo -o p -- python $R/tools/bench_ppo.py --iters 2 > $R/gpurun_out/r06_prof_ppo.log 2>&1 )
      f=$(find gpurun_out/r06_prof_ppo -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r06_ppo_kernel_stats.csv; head -45 gpurun_out/r06_ppo_kernel_stats.csv | cut -c1-170; tail -2 gpurun_out/r06_prof_ppo.log | cut -c1-900
      find gpurun_out/r06_prof_ppo -name "*kernel_trace.csv" -delete ;;
    decode_ab)       # round-6 decode launch rules (8 waves per strip when a narrow launch has more strips than CUs; four key steps in flight in the cache attention at one or two sequences) against round 5's (AA_DECODE_R6=0): numerics, then the PPO iteration, alternating
      timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_llama3_gpu.py tests/test_ppo_gpu.py tests/test_grpo_gpu.py -q -x -m gpu -p no:cacheprovider > gpurun_out/r06_decode_ab_tests.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r06_decode_ab_tests.log | cut -c1-300
      for v in 0 1 0 1; do
        AA_DECODE_R6=$v timeout 400 python tools/bench_ppo.py --iters 2 > gpurun_out/r06_bench_ppo_r6_$v.json 2> gpurun_out/r06_bench_ppo_r6_$v.err
        python -c "import json; d=json.loads([l for l in open('gpurun_out/r06_bench_ppo_r6_$v.json') if l.startswith('{')][-1]); print('AA_DECODE_R6=$v iteration', round(d['iteration_ms'],1), 'ms', {k: round(x,2) for k,x in d['split_ms'].items()}, {k: d[k] for k in d if 'position' in k})" || tail -3 gpurun_out/r06_bench_ppo_r6_$v.err
      done ;;
    decode_ab2)      # the further decode rules of round 6 (bit 1: pipelined deep strips; bit 2: eight key steps in flight): numerics, then the PPO iteration under masks 1 / 3 / 5 / 7, twice around
      timeout 600 python -m pytest tests/test_decode_gpu.py -q -x -m gpu -p no:cacheprovider > gpurun_out/r06_decode_ab2_tests.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r06_decode_ab2_tests.log | cut -c1-300
      for v in ${AA_AB2_MASKS:-1 3 5 7 1 3 5 7}; do
        AA_DECODE_R6=$v timeout 400 python tools/bench_ppo.py --iters 2 > gpurun_out/r06_bench_ppo_r6_$v.json 2> gpurun_out/r06_bench_ppo_r6_$v.err
        python -c "import json; d=json.loads([l for l in open('gpurun_out/r06_bench_ppo_r6_$v.json') if l.startswith('{')][-1]); print('AA_DECODE_R6=$v iteration', round(d['iteration_ms'],1), 'ms', {k: round(x,2) for k,x in d['split_ms'].items()}, {k: d[k] for k in d if 'position' in k})" || tail -3 gpurun_out/r06_bench_ppo_r6_$v.err
      done ;;
    decode_ab3)      # the strip kernel's epilogue operands (bias / residual; position -> cos / sin, cache slot) requested above the weight stream, against the build before it (libaa_hip_nohoist.so), both under rules 3: numerics, then the PPO iteration, alternating
      timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_llama3_gpu.py tests/test_ppo_gpu.py tests/test_grpo_gpu.py tests/test_qwen2vl_gpu.py tests/test_qwen3moe_gpu.py -q -x -m gpu -p no:cacheprovider -k "not width_pair" > gpurun_out/r06_decode_ab3_tests.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r06_decode_ab3_tests.log | cut -c1-300
      for v in libaa_hip_nohoist.so libaa_hip.so libaa_hip_nohoist.so libaa_hip.so; do
        AA_DECODE_R6=3 AA_HIP_LIB=$R/align_anything_amd/$v timeout 400 python tools/bench_ppo.py --iters 2 > gpurun_out/r06_bench_ppo_$v.json 2> gpurun_out/r06_bench_ppo_$v.err
        python -c "import json; d=json.loads([l for l in open('gpurun_out/r06_bench_ppo_$v.json') if l.startswith('{')][-1]); print('$v iteration', round(d['iteration_ms'],1), 'ms', {k: round(x,2) for k,x in d['split_ms'].items()}, {k: d[k] for k in d if 'position' in k})" || tail -3 gpurun_out/r06_bench_ppo_$v.err
      done ;;
    decode_batches_ab)   # rule bit 2 (four key steps in flight whenever H x N < 512) at the batches between PPO's 1 and the 4-wave form: 4 / 10 (GRPO: B x num_generations) / 16 sequences, masks 3 and 7 alternating
      timeout 300 python -m pytest tests/test_decode_gpu.py -q -x -m gpu -p no:cacheprovider -k "attention or launch_rules" 2>&1 | tail -2 | cut -c1-200
      for v in 3 7 3 7; do
        AA_DECODE_R6=$v AA_BENCH_DECODE_CASES="4,512,64;10,512,64;16,512,64" AA_BENCH_DECODE_OUT=r06_bench_decode_rules$v.json timeout 600 python tools/bench_decode.py 2>&1 | grep -E "^\{" | python -c "
import sys, ast
print('AA_DECODE_R6=$v', [(r['N'], round(r['ms_per_step'], 4), round(r['tokens_per_s'])) for r in map(ast.literal_eval, sys.stdin)])"
      done ;;
    rocprof_decode)  # rocprofv3 --kernel-trace --stats of the decode loop at 1 and at 10 sequences (GRPO's B x num_generations): which kernels grow with the batch
      for n in 1 10; do
        ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r06_prof_dec$n && AA_BENCH_DECODE_CASES="$n,512,128" AA_BENCH_DECODE_OUT=r06_bench_decode_prof$n.json timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_prof_dec$n -o p -- python $R/tools/bench_decode.py > $R/gpurun_out/r06_prof_dec$n.log 2>&1 )
        f=$(find gpurun_out/r06_prof_dec$n -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r06_decode_kernel_stats_n$n.csv; echo "--- $n sequence(s)"; head -12 gpurun_out/r06_decode_kernel_stats_n$n.csv | cut -c1-60,200-330
        find gpurun_out/r06_prof_dec$n -name "*kernel_trace.csv" -delete
      done ;;
    dp2)             # smoke(), then the N = 2 code path of bench.py as a FUNCTIONAL run on one device (gloo; never a performance number) on the round's final code
      timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
      AA_BENCH_ONE_DEVICE=1 AA_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 2 --warmup 1 --layers 4 --no-cpu-baseline --no-gemm-events > gpurun_out/r06_dp2_functional_onebox.json 2> gpurun_out/r06_dp2_functional_onebox.err; echo "rc=$?"
      python -c "import json; d=json.loads(open('gpurun_out/r06_dp2_functional_onebox.json').read().strip().split(chr(10))[-1]); m=d['multi_gpu']; print('dp2 functional:', d['config']['workload'][-60:], 'n_gpus', d['n_gpus'], 'value', round(d['value'], 3), 'replicas identical', m['replicas_bit_identical_after_steps'], 'reduce', m['reduce_mode'], (m.get('reduce_autotune') or {}).get('forms_agree'))" || tail -8 gpurun_out/r06_dp2_functional_onebox.err ;;
    *) echo "unknown stage $stage" ;;
  esac
done
