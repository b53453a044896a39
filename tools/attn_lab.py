"""Attention kernel lab: same-box A/B of builds of csrc/attention.hip (tools/build_attn_variant.sh) at the benchmarked geometry.

    python tools/attn_lab.py                       parent: runs itself once per library in AA_ATTN_LIBS (comma list of
                                                   libaa_hip*.so names under align_anything_amd/, first = the baseline),
                                                   prints the timings and whether every output is BIT-IDENTICAL to the baseline's
    AA_HIP_LIB=... python tools/attn_lab.py child  one library: hashes + timings as one JSON line

Cases: the bench's causal 8 x 32 heads x 2048 x 128 block (timed), the same with two left-padded rows, a GQA ragged case,
the CLIP tower's non-causal 577 x 64 heads and a right-padded (kv_len) encoder case.  A scheduling-only change of the kernels must
reproduce the baseline's bits exactly.
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from align_anything_amd import ops
    dev = torch.device('cuda:0')

    def h(*ts):
        m = hashlib.sha1()
        for t in ts:
            m.update(t.detach().contiguous().cpu().view(torch.uint8).numpy().tobytes())
        return m.hexdigest()[:16]

    def case(N, T, H, Hkv, hd, causal, start=None, kv_len=None, seed=0, time_it=False):
        g = torch.Generator(device='cpu').manual_seed(seed)
        W = (H + 2 * Hkv) * hd
        qkv = (torch.randn(N * T, W, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        q, k, v = qkv[:, :H * hd], qkv[:, H * hd:(H + Hkv) * hd], qkv[:, (H + Hkv) * hd:]
        do = (torch.randn(N * T, H * hd, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        st = None if start is None else torch.tensor(start, dtype=torch.int32, device=dev)
        kl = None if kv_len is None else torch.tensor(kv_len, dtype=torch.int32, device=dev)
        sc = hd ** -0.5
        o, lse = ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, causal, sc, start=st, kv_len=kl)
        dqkv = torch.zeros_like(qkv)
        dq, dk, dv = dqkv[:, :H * hd], dqkv[:, H * hd:(H + Hkv) * hd], dqkv[:, (H + Hkv) * hd:]
        ops.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, N, T, H, Hkv, hd, causal, sc, start=st, kv_len=kl)
        torch.cuda.synchronize()
        out = {'fwd': h(o, torch.nan_to_num(lse, neginf=-1e30)), 'bwd': h(dqkv), 'finite': bool(torch.isfinite(dqkv.float()).all())}
        if time_it:
            reps = 20
            for name, fn in (('fwd', lambda: ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, causal, sc, start=st, kv_len=kl, out=o)),
                             ('bwd', lambda: ops.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, N, T, H, Hkv, hd, causal, sc, start=st, kv_len=kl))):
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                best = 1e9
                for _ in range(3):
                    e0.record()
                    for _ in range(reps):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / reps)
                fl = 4.0 * N * H * T * T * hd * (0.5 if causal else 1.0) * (1.0 if name == 'fwd' else 2.5)
                out[name + '_us'] = round(best * 1e3, 1)
                out[name + '_tflops'] = round(fl / best / 1e9, 1)
        return out

    if os.environ.get('AA_LAB_ONLY') == 'bench':        # under rocprofv3: only the timed block, so the per-kernel averages are its own
        case(8, 2048, 32, 32, 128, True, time_it=True)
        return
    if os.environ.get('AA_LAB_CASES') == 'bench':       # timing-only lab builds (AA_BWD_LAB): the bench block alone
        print('ATTNLAB ' + json.dumps({'bench': case(8, 2048, 32, 32, 128, True, time_it=True)}), flush=True)
        return
    res = {
        'bench': case(8, 2048, 32, 32, 128, True, time_it=True),
        'leftpad': case(8, 2048, 32, 32, 128, True, start=[0, 700, 0, 0, 1531, 0, 0, 64], seed=1),
        'gqa_ragged': case(2, 1000, 8, 2, 128, True, start=[0, 333], seed=2),
        'clip': case(4, 577, 16, 16, 64, False, seed=3, time_it=True),
        'kvlen': case(3, 750, 20, 20, 64, False, kv_len=[750, 401, 17], seed=4),
        'hd64_causal': case(2, 512, 12, 12, 64, True, start=[5, 0], seed=5),
    }
    print('ATTNLAB ' + json.dumps(res), flush=True)


def parent():
    libs = os.environ.get('AA_ATTN_LIBS', 'libaa_hip_old.so,libaa_hip.so').split(',')
    base = None
    rows = {}
    for lib in libs:
        path = os.path.join(ROOT, 'align_anything_amd', lib)
        if not os.path.exists(path):
            print(f'{lib}: missing'); continue
        env = dict(os.environ, AA_HIP_LIB=path)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('ATTNLAB ')]
        if not line:
            print(f'{lib}: FAILED rc={r.returncode}\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}'); continue
        res = json.loads(line[0][8:])
        rows[lib] = res
        if base is None:
            base = res
        same = {c: (res[c]['fwd'] == base[c]['fwd'], res[c]['bwd'] == base[c]['bwd']) for c in res}
        b = res['bench']; c = res.get('clip', {'fwd_us': 0.0, 'bwd_us': 0.0})
        print(f"{lib:24s} bench fwd {b['fwd_us']:7.1f} us {b['fwd_tflops']:6.1f} TF | bwd {b['bwd_us']:7.1f} us {b['bwd_tflops']:6.1f} TF | clip fwd {c['fwd_us']:6.1f} bwd {c['bwd_us']:6.1f} us"
              f" | identical to {libs[0]}: " + ' '.join(f"{k}={'ok' if all(v) else 'DIFF' + str(v)}" for k, v in same.items())
              + ('' if all(res[k]['finite'] for k in res) else ' NON-FINITE'), flush=True)
    out = os.environ.get('AA_LAB_OUT')
    if out:
        with open(os.path.join(ROOT, 'gpurun_out', out), 'w') as f:
            json.dump(rows, f, indent=1)


if __name__ == '__main__':
    child() if len(sys.argv) > 1 and sys.argv[1] == 'child' else parent()
