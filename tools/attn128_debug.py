"""Debug probes for csrc/attn128.inc (structured inputs that expose the operand mappings)."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from align_anything_amd import ops
from align_anything_amd.lib import LIB

dev = torch.device('cuda:0')
LIB.load()
hd = 128
torch.set_printoptions(linewidth=250, precision=4, sci_mode=False)

def run(q, k, v, N, T, H, Hkv, causal, impl, start=None):
    LIB.call('aa_attn_set_impl', impl)
    o, lse = ops.attn_fwd(q.contiguous(), k.contiguous(), v.contiguous(), N, T, H, Hkv, hd, causal, hd ** -0.5, start=start)
    torch.cuda.synchronize()
    return o.float(), lse

for T in (64, 128, 256):
    N, H = 1, 1
    g = torch.Generator().manual_seed(0)
    # A: K = 0 -> uniform P
    q = (torch.randn(T, hd, generator=g) * 0.5).bfloat16().to(dev)
    k = torch.zeros(T, hd).bfloat16().to(dev)
    v = (torch.randn(T, hd, generator=g)).bfloat16().to(dev)
    o, lse = run(q, k, v, N, T, H, H, False, 3)
    print(f'T={T} A uniform: lse min/max', float(lse.min()), float(lse.max()), 'want', float(torch.log(torch.tensor(float(T)))), ' max|O-mean(V)|', float((o - v.float().mean(0, keepdim=True)).abs().max()))
    # B: V one-hot on (key % 128) -> O[q][d] = sum_{key = d mod 128} P[q][key]
    k = (torch.randn(T, hd, generator=g) * 0.5).bfloat16().to(dev)
    v = torch.zeros(T, hd)
    v[torch.arange(T), torch.arange(T) % hd] = 1.0
    v = v.bfloat16().to(dev)
    o, lse = run(q, k, v, N, T, H, H, False, 3)
    s = (q.float() @ k.float().T) * hd ** -0.5
    p = torch.softmax(s, -1)
    want = torch.zeros(T, hd, device=dev)
    want.index_add_(1, torch.arange(T, device=dev) % hd, p)
    err = (o - want).abs()
    print(f'T={T} B one-hot V: max err', float(err.max()), 'lse err', float((lse.view(-1) - torch.logsumexp(s, -1)).abs().max()))
    bad = (err > 2e-3).nonzero()
    print('   bad entries', bad.shape[0], 'rows with bad', torch.unique(bad[:, 0]).numel(), 'cols with bad', torch.unique(bad[:, 1])[:40].tolist())
    if bad.shape[0]:
        r = int(bad[0, 0])
        print('   row', r, 'got ', o[r, :min(T, 64)].tolist()[:48])
        print('   row', r, 'want', want[r, :min(T, 64)].tolist()[:48])
    # C: scores probe: V = 0 except V[key j][0] = 1 for one j at a time is slow; instead lse with a single hot key
    for hot in (0, 5, 31, 32, 40, 63):
        if hot >= T: continue
        k2 = torch.zeros(T, hd); k2[hot] = 4.0
        q2 = torch.full((T, hd), 0.25)
        o2, lse2 = run(q2.bfloat16().to(dev), k2.bfloat16().to(dev), v, N, T, H, H, False, 3)
        s2 = (q2 @ k2.T) * hd ** -0.5
        want_lse = torch.logsumexp(s2, -1)
        print(f'   hot key {hot}: lse got {float(lse2.view(-1)[0]):.4f} .. {float(lse2.view(-1)[-1]):.4f} want {float(want_lse[0]):.4f}; O[0][hot%128] = {float(o2[0, hot % hd]):.4f} (want ~{float(torch.softmax(s2, -1)[0, hot]):.4f})')
