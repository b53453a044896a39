#!/bin/bash
# secondary datapoints on the round's kernels: rollout decode at 7B, DPO step on the Qwen2-VL / Qwen2-Audio / Qwen3-MoE geometries
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD
AA_BENCH_DECODE_QUICK=1 AA_BENCH_DECODE_AB=1 timeout 300 python tools/bench_decode.py 2>/dev/null | tail -3 | tee gpurun_out/r02_bench_decode_7b.json | cut -c1-400
timeout 300 python tools/bench_qwen2vl.py 2>/dev/null | tail -1 | tee gpurun_out/r02_bench_qwen2vl_7b_dpo.json | cut -c1-400
timeout 300 python tools/bench_qwen2audio.py 2>/dev/null | tail -1 | tee gpurun_out/r02_bench_qwen2audio_7b_dpo.json | cut -c1-400
timeout 300 python tools/bench_qwen3moe.py 2>/dev/null | tail -1 | tee gpurun_out/r02_bench_qwen3moe_12layers_dpo.json | cut -c1-400
