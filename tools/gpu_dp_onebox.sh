#!/bin/bash
# Functional check of bench.py's N=2 path on a single-GPU box (ranks share cuda:0, gloo collectives).
set -x
export AA_BENCH_ONE_DEVICE=1 AA_BENCH_BACKEND=gloo
mkdir -p gpurun_out
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 3 --warmup 1 --layers 4 --pairs-per-gpu 1 --no-cpu-baseline > gpurun_out/dp2_onebox.log 2>&1
echo rc=$?
tail -5 gpurun_out/dp2_onebox.log
unset AA_BENCH_ONE_DEVICE AA_BENCH_BACKEND
timeout 300 python bench.py --steps 3 --warmup 1 --layers 4 --pairs-per-gpu 1 --no-cpu-baseline 2>&1 | tail -2
