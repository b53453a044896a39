"""Per-step / per-barrier s_memtime trace of ONE workgroup of csrc/attn128.inc (-DA128_TRACE lab build): AA_HIP_LIB=.../libaa_hip_trace.so python tools/attn128_trace.py"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from align_anything_amd import ops
from align_anything_amd.lib import LIB
dev = torch.device('cuda:0'); LIB.load(); hd = 128
for name, (N, T, H, Hkv, causal) in {'guide_noncausal': (16, 2048, 64, 8, False), 'bench_causal': (8, 2048, 32, 32, True)}.items():
    g = torch.Generator().manual_seed(3)
    qkv = (torch.randn(N * T, (H + 2 * Hkv) * hd, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:(H + Hkv) * hd], qkv[:, (H + Hkv) * hd:]
    LIB.call('aa_attn_set_impl', 3)
    for _ in range(2):
        o, lse = ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, causal, hd ** -0.5)
    torch.cuda.synchronize()
    raw = (lse.view(torch.int32).cpu().numpy().reshape(-1)[T - 256:T].astype(np.int64) & 0xffffffff).reshape(4, 64)
    ntile = 16
    print(name)
    for w in range(4):
        t = raw[w, :4 * ntile].reshape(ntile, 4)          # per tile: step<0> start, after step<0>, after vmcnt wait, after barrier
        d0 = (t[:, 1] - t[:, 0]) & 0xffffffff             # even step
        dw = (t[:, 2] - t[:, 1]) & 0xffffffff             # vmcnt wait
        db = (t[:, 3] - t[:, 2]) & 0xffffffff             # barrier
        nxt = np.roll(t[:, 0], -1)
        d1 = (nxt - t[:, 3]) & 0xffffffff                 # odd step (+ dm setup)
        sl = slice(0, ntile - 1)
        print(f'  wave {w}: even step {d0[sl].mean():7.0f} (min {d0[sl].min()} max {d0[sl].max()})  vmcnt wait {dw[sl].mean():6.0f} (max {dw[sl].max()})  barrier {db[sl].mean():6.0f} (max {db[sl].max()})  odd step {d1[sl].mean():7.0f} (min {d1[sl].min()} max {d1[sl].max()})')
        if w == 0:
            print('     even', d0[2:12].tolist()); print('     odd ', d1[2:12].tolist()); print('     bar ', db[2:12].tolist()); print('     vm  ', dw[2:12].tolist())
