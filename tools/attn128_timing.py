"""Block-phase cycle anatomy of csrc/attn128.inc from an -DA128_TIMING lab build (tools/build_attn_variant.sh timing -DA128_TIMING):
    AA_HIP_LIB=.../libaa_hip_timing.so python tools/attn128_timing.py
Every wave writes s_memtime stamps (entry, Q prescaled / DMA issued, loop entry, loop exit, stores drained) over the first lse entries of its rows."""
import os, sys, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from align_anything_amd import ops
from align_anything_amd.lib import LIB

dev = torch.device('cuda:0')
LIB.load()
hd = 128
SHAPES = {'bench_causal': (8, 2048, 32, 32, True), 'guide_noncausal': (16, 2048, 64, 8, False)}
if os.environ.get('A128_TIMING_ONLY'):
    SHAPES = {k: v for k, v in SHAPES.items() if k in os.environ['A128_TIMING_ONLY'].split(',')}
for name, (N, T, H, Hkv, causal) in SHAPES.items():
    g = torch.Generator().manual_seed(3)
    qkv = (torch.randn(N * T, (H + 2 * Hkv) * hd, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:(H + Hkv) * hd], qkv[:, (H + Hkv) * hd:]
    LIB.call('aa_attn_set_impl', 3)
    for _ in range(3):
        o, lse = ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, causal, hd ** -0.5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); o, lse = ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, causal, hd ** -0.5); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    raw = lse.view(torch.int32).cpu().numpy().reshape(N * H, T).astype(np.int64) & 0xffffffff
    w = raw.reshape(N * H, T // 64, 64)
    ok = w[:, :, 0] == 0x7157a3b1
    st = w[:, :, 1:6]
    d = (st[:, :, 1:] - st[:, :, :-1]) & 0xffffffff           # prologue a, prologue b, loop, epilogue
    nh = w[:, :, 6]; ntile = w[:, :, 7]
    hi = w[:, :, 8]
    t0 = st[:, :, 0] + (hi << 32); t4 = t0 + ((st[:, :, 4] - st[:, :, 0]) & 0xffffffff)
    span = (t4.max() - t0.min())
    print(f'{name}: kernel {us:.1f} us; stamped waves {int(ok.sum())}/{ok.size}; s_memtime span {span} ticks -> {span / us:.1f} ticks/us')
    names = ['entry->Q prescaled+DMA issued', 'wait first tiles + H0 scores', 'main loop', 'epilogue (normalize+store+drain)']
    tot = d.sum(-1)
    for i, nm in enumerate(names):
        print(f'   {nm:36s} mean {d[:, :, i].mean():9.0f}  p50 {np.median(d[:, :, i]):9.0f}  max {d[:, :, i].max():9.0f}   {100 * d[:, :, i].sum() / tot.sum():5.1f} % of wave time')
    steps = np.maximum(2 * ntile + 1, 1)                      # steps a wave is resident for (it waits at the tile barriers of the others)
    per = d[:, :, 2] / steps
    print(f'   loop ticks per step (block-resident steps 2 ntile + 1): mean {per.mean():.0f}  p10 {np.percentile(per, 10):.0f}  p90 {np.percentile(per, 90):.0f}')
    for wv in range(4):
        sel = per.reshape(N * H, T // 256, 4)[:, :, wv]
        print(f'      wave {wv}: {sel.mean():.0f} ticks/step, own half tiles {nh.reshape(N * H, T // 256, 4)[:, :, wv].mean():.1f} of {2 * ntile.reshape(N * H, T // 256, 4)[:, :, wv].mean():.1f}')
    # busy fraction of the chip: sum of block wave-time / (CUs * span)
    blk = tot.reshape(N * H, T // 256, 4).max(-1)
    print(f'   sum of block times / (256 CUs x span) = {blk.sum() / (256 * span):.3f}')
    hw = w[:, :, 9].reshape(N * H, T // 256, 4)[:, :, 0]
    cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 16) & 0xf) << 7)         # cu_id, se_id, xcc? (diagnostic only)
    print(f'   distinct (cu, se, ..) ids seen: {len(np.unique(cu))}')
