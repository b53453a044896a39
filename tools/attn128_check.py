"""Lab / acceptance check of the head_dim-128 attention kernels of csrc/attn128.inc against (a) an fp32 torch softmax reference and (b) the 16x16x32 kernels of the
same library (aa_attn_set_impl), with timings of both on the bench block (causal MHA 8 x 32 x 2048 x 128) and on the guide's non-causal GQA block.

    python tools/attn128_check.py [--bwd] [--no-time]        -> gpurun_out/attn128_check.json, one JSON line per case on stdout

Cases: causal bench block, left padding (incl. a fully padded row block), GQA with ragged T, right padding (kv_len), tiny T (< one tile, one half tile + 1), a spiked
key that forces the deferred running maximum to move late (cdna_hip_programming.md T13's test rule)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from align_anything_amd import ops  # noqa: E402
from align_anything_amd.lib import LIB  # noqa: E402


def ref(q, k, v, do, N, T, H, Hkv, hd, causal, sc, start, kv_len, want_bwd):
    dev = q.device
    qf = q.float().view(N, T, H, hd).transpose(1, 2).detach().requires_grad_(want_bwd)
    kf = k.float().view(N, T, Hkv, hd).transpose(1, 2).detach().requires_grad_(want_bwd)
    vf = v.float().view(N, T, Hkv, hd).transpose(1, 2).detach().requires_grad_(want_bwd)
    g = H // Hkv
    s = (qf @ kf.repeat_interleave(g, 1).transpose(-1, -2)) * sc
    idx = torch.arange(T, device=dev)
    ok = torch.ones(N, 1, T, T, dtype=torch.bool, device=dev)
    if causal:
        ok = ok & (idx[None, :] <= idx[:, None])[None, None]
    if start is not None:
        ok = ok & (idx[None, None, None, :] >= start.long()[:, None, None, None])
    if kv_len is not None:
        ok = ok & (idx[None, None, None, :] < kv_len.long()[:, None, None, None])
    s = s.masked_fill(~ok, float('-inf'))
    lse = torch.logsumexp(s, -1)
    p = torch.exp(s - torch.where(torch.isfinite(lse), lse, torch.zeros_like(lse))[..., None])
    p = torch.where(ok, p, torch.zeros_like(p))
    o = p @ vf.repeat_interleave(g, 1)
    back = lambda t, h: t.transpose(1, 2).reshape(N * T, h * hd)
    out = {'o': back(o.detach(), H), 'lse': lse.detach()}
    if want_bwd:
        (o * do.float().view(N, T, H, hd).transpose(1, 2)).sum().backward()
        out.update(dq=back(qf.grad, H), dk=back(kf.grad, Hkv), dv=back(vf.grad, Hkv))
    return out


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bwd', action='store_true')
    ap.add_argument('--no-time', action='store_true')
    ap.add_argument('--impl', type=int, default=3, help='aa_attn_set_impl value of the kernels under test')
    ap.add_argument('--base', type=int, default=0, help='aa_attn_set_impl value of the kernels compared against (0 = the 16x16x32 kernels; 3 + --impl 7: the dK/dV variants)')
    ap.add_argument('--only', default='', help='comma list of case names')
    ap.add_argument('--out', default='attn128_check.json')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    LIB.load()
    hd = 128
    cases = {
        'bench': (8, 2048, 32, 32, True, None, None, True),
        'leftpad': (4, 2048, 8, 8, True, [0, 37, 700, 2047], None, False),
        'leftpad_blocks': (2, 1024, 4, 4, True, [256, 511], None, False),
        'gqa_ragged': (3, 1000, 28, 4, True, [0, 5, 130], None, False),
        'kvlen': (2, 1500, 8, 8, False, None, [1500, 777], False),
        'kvlen_causal': (2, 700, 4, 2, True, [0, 3], [700, 650], False),
        'short100': (2, 100, 4, 4, True, None, None, False),
        'short33': (2, 33, 4, 2, True, [0, 2], None, False),
        'short64_nc': (1, 64, 2, 2, False, None, None, False),
        'spike': (1, 1024, 2, 2, True, None, None, False),
        'guide_noncausal_gqa': (16, 2048, 64, 8, False, None, None, True),
    }
    res = []
    for name, (N, T, H, Hkv, causal, start, kv_len, timed) in cases.items():
        if a.only and name not in a.only.split(','):
            continue
        g = torch.Generator().manual_seed(1 + len(name))
        qkv = (torch.randn(N * T, (H + 2 * Hkv) * hd, generator=g) * 0.5).to(torch.bfloat16)
        if name == 'spike':     # key 900 of head 0 lines up with queries >= 900: their maxima jump by ~40 in the log2 domain at tile 14, long after tile 0
            qkv[900:, :hd] = 0.25
            qkv[900, H * hd:H * hd + hd] = 8.0
        qkv = qkv.to(dev)
        q, k, v = qkv[:, :H * hd], qkv[:, H * hd:(H + Hkv) * hd], qkv[:, (H + Hkv) * hd:]
        do = (torch.randn(N * T, H * hd, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        st = None if start is None else torch.tensor(start, dtype=torch.int32, device=dev)
        kl = None if kv_len is None else torch.tensor(kv_len, dtype=torch.int32, device=dev)
        sc = hd ** -0.5
        c = {'case': name, 'shape': [N, T, H, Hkv], 'causal': causal}
        outs = {}
        for impl in (a.base, a.impl):
            LIB.call('aa_attn_set_impl', impl)
            o, lse = ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, causal, sc, start=st, kv_len=kl)
            d = {'o': o, 'lse': lse}
            if a.bwd:
                dqkv = torch.zeros_like(qkv)
                dq, dk, dv = dqkv[:, :H * hd], dqkv[:, H * hd:(H + Hkv) * hd], dqkv[:, (H + Hkv) * hd:]
                ops.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, N, T, H, Hkv, hd, causal, sc, start=st, kv_len=kl)
                d.update(dq=dq, dk=dk, dv=dv)
            torch.cuda.synchronize()
            outs[impl] = d
        new, old = outs[a.impl], outs[a.base]
        c['o_bit_identical_to_base'] = bool(torch.equal(new['o'], old['o'])) and bool(torch.equal(new['lse'], old['lse']))
        if a.bwd:
            for t_ in ('dq', 'dk', 'dv'):
                c['%s_bit_identical_to_base' % t_] = bool(torch.equal(new[t_], old[t_]))
        small = N * H * T * T <= 8 * 32 * 2048 * 2048
        if small:
            r = ref(q, k, v, do, N, T, H, Hkv, hd, causal, sc, st, kl, a.bwd)
            fin = torch.isfinite(r['lse'])
            # rows that can attend nothing (left padding at or beyond the row): O = 0, lse = -inf in both
            valid_rows = fin.transpose(1, 2).reshape(N * T, H).repeat_interleave(hd, 1)
            c['rel_o_vs_ref'] = rel(new['o'] * valid_rows, r['o'] * valid_rows)
            c['rel_o_old_vs_ref'] = rel(old['o'] * valid_rows, r['o'] * valid_rows)
            c['max_abs_o_vs_ref'] = float(((new['o'].float() - r['o']) * valid_rows).abs().max())
            c['masked_rows_zero'] = bool((new['o'].float() * (~valid_rows)).abs().max() == 0)
            c['lse_inf_pattern'] = bool(torch.equal(torch.isfinite(new['lse']), fin))
            c['max_abs_lse'] = float((new['lse'][fin] - r['lse'][fin]).abs().max()) if fin.any() else 0.0
            c['ok'] = c['rel_o_vs_ref'] < 6e-3 and c['masked_rows_zero'] and c['lse_inf_pattern'] and c['max_abs_lse'] < 2e-3 and \
                bool(torch.isfinite(new['o'].float()).all())
            if a.bwd:
                for t_ in ('dq', 'dk', 'dv'):
                    c['rel_%s_vs_ref' % t_] = rel(new[t_], r[t_])
                    c['rel_%s_old_vs_ref' % t_] = rel(old[t_], r[t_])
                    c['ok'] = c['ok'] and c['rel_%s_vs_ref' % t_] < 1.2e-2 and bool(torch.isfinite(new[t_].float()).all())
            del r
        else:
            c['rel_o_vs_old'] = rel(new['o'], old['o'])
            c['ok'] = c['rel_o_vs_old'] < 6e-3
        if timed and not a.no_time:
            fl = 4.0 * N * H * T * T * hd * (0.5 if causal else 1.0)
            for impl in (a.base, a.impl, a.base, a.impl):
                LIB.call('aa_attn_set_impl', impl)
                o = outs[impl]['o']
                fns = [('fwd', 1.0, lambda: ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, causal, sc, start=st, kv_len=kl, out=o))]
                if a.bwd:
                    d = outs[impl]
                    fns.append(('bwd', 2.5, lambda: ops.attn_bwd(q, k, v, d['o'], do, d['lse'], d['dq'], d['dk'], d['dv'], N, T, H, Hkv, hd, causal, sc, start=st, kv_len=kl)))
                for tag, mult, fn in fns:
                    for _ in range(3):
                        fn()
                    best = 1e9
                    for _ in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(10):
                            fn()
                        e1.record()
                        torch.cuda.synchronize()
                        best = min(best, e0.elapsed_time(e1) / 10)
                    key = f'{tag}_impl{impl}'
                    c.setdefault(key + '_us', []).append(round(best * 1e3, 1))
                    c.setdefault(key + '_tflops', []).append(round(fl * mult / (best * 1e-3) / 1e12, 1))
        res.append(c)
        print(json.dumps(c), flush=True)
        del outs, qkv, q, k, v, do
        torch.cuda.empty_cache()
    LIB.call('aa_attn_set_impl', 7)
    print('ALL OK' if all(c['ok'] for c in res) else 'MISMATCH: ' + ', '.join(c['case'] for c in res if not c['ok']), flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', a.out), 'w') as f:
        json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
