"""Attention kernels at the geometry the CDNA4 guide quotes its numbers on (cdna_hip_programming.md, "Fused attention prefill": GQA B = 16, H = 64, H_KV = 8,
N = 2048, D = 128, random data, non-causal: plain-HIP ladder ~1000 TFLOP/s, tuned 8-wave ~1200, one-wave-per-SIMD asm structure 1250-1400), next to the bench's own
causal 8 x 32 x 2048 x 128 block -- so that the distance of csrc/attention.hip from those levels is a measured number, not an estimate.

    python tools/attn_guide_geometry.py > gpurun_out/r03_attention_guide_geometry.json
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align_anything_amd import ops  # noqa: E402


def run(N, T, H, Hkv, hd, causal, reps=20):
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    W = (H + 2 * Hkv) * hd
    qkv = (torch.randn(N * T, W, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:(H + Hkv) * hd], qkv[:, (H + Hkv) * hd:]
    do = (torch.randn(N * T, H * hd, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    sc = hd ** -0.5
    o, lse = ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, causal, sc)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[:, :H * hd], dqkv[:, H * hd:(H + Hkv) * hd], dqkv[:, (H + Hkv) * hd:]
    out = {'N': N, 'T': T, 'H': H, 'Hkv': Hkv, 'hd': hd, 'causal': causal}
    flop_fwd = 4.0 * N * H * T * T * hd * (0.5 if causal else 1.0)
    for name, fn, fl in (('fwd', lambda: ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, causal, sc, out=o), flop_fwd),
                         ('bwd', lambda: ops.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, N, T, H, Hkv, hd, causal, sc), 2.5 * flop_fwd)):
        for _ in range(3):
            fn()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        out[name + '_us'] = round(best * 1e3, 1)
        out[name + '_tflops'] = round(fl / (best * 1e-3) / 1e12, 1)
        out[name + '_frac_of_2500'] = round(fl / (best * 1e-3) / 2.5e15, 4)
    return out


if __name__ == '__main__':
    res = {'what': 'csrc/attention.hip, bf16, random data; TFLOP/s on algorithmic flops (causal counted at half; backward = 2.5 x forward)',
           'guide_levels_noncausal_tflops': {'plain_hip_ladder': 1000, 'tuned_8_wave': 1200, 'one_wave_per_simd_asm': [1250, 1400]},
           'cases': [run(16, 2048, 64, 8, 128, False), run(16, 2048, 64, 8, 128, True), run(8, 2048, 32, 32, 128, True), run(8, 2048, 32, 32, 128, False)]}
    print(json.dumps(res))
