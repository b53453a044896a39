"""Lab: aa_attn_fwd_v2 (tools/lab/attn_v2/attention_v2.hip, 32 x 32 x 16 MFMA, 64 query rows per wave) against the shipped aa_attn_fwd of the SAME lab library.

    bash tools/lab/attn_v2/build.sh && AA_HIP_LIB=$PWD/align_anything_amd/libaa_hip_v2.so python tools/lab/attn_v2/check.py

Numerics: O within bf16 rounding of the shipped kernel's (the 32 x 32 tiles add the same products in another order: not bit-identical), lse to 1e-3; cases as in
tools/attn_lab.py (causal bench block, left padding, GQA ragged T, right padding via kv_len, non-causal).  Timing: the bench block and the guide's non-causal GQA block."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from align_anything_amd import ops  # noqa: E402
from align_anything_amd.lib import LIB  # noqa: E402


def main():
    dll = LIB.load()
    v2 = dll.aa_attn_fwd_v2
    v2.restype = ctypes.c_int
    v2.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_long] * 4 + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_void_p]
    dev = torch.device('cuda:0')

    def fwd_v2(q, k, v, N, T, H, Hkv, hd, causal, sc, start=None, kv_len=None, out=None):
        o = torch.empty((N * T, H * hd), dtype=q.dtype, device=dev) if out is None else out
        lse = torch.empty((N, H, T), dtype=torch.float32, device=dev)
        rc = v2(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), start.data_ptr() if start is not None else None,
                kv_len.data_ptr() if kv_len is not None else None, q.stride(0), k.stride(0), v.stride(0), o.stride(0), N, T, H, Hkv, hd, int(causal), float(sc),
                ops.stream())
        if rc:
            raise RuntimeError(dll.aa_last_error().decode())
        return o, lse

    res = {'lib': os.environ.get('AA_HIP_LIB'), 'cases': []}
    for name, (N, T, H, Hkv, causal, start, kv_len, timed) in {
            'bench': (8, 2048, 32, 32, True, None, None, True),
            'leftpad': (4, 2048, 8, 8, True, [0, 37, 700, 2047], None, False),
            'gqa_ragged': (3, 1000, 28, 4, True, [0, 5, 130], None, False),
            'kvlen': (2, 1500, 8, 8, False, None, [1500, 777], False),
            'short': (2, 100, 4, 4, True, None, None, False),
            'guide_noncausal_gqa': (16, 2048, 64, 8, False, None, None, True)}.items():
        hd = 128
        g = torch.Generator().manual_seed(1)
        qkv = (torch.randn(N * T, (H + 2 * Hkv) * hd, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        q, k, v = qkv[:, :H * hd], qkv[:, H * hd:(H + Hkv) * hd], qkv[:, (H + Hkv) * hd:]
        st = None if start is None else torch.tensor(start, dtype=torch.int32, device=dev)
        kl = None if kv_len is None else torch.tensor(kv_len, dtype=torch.int32, device=dev)
        sc = hd ** -0.5
        o1, l1 = ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, causal, sc, start=st, kv_len=kl)
        o2, l2 = fwd_v2(q, k, v, N, T, H, Hkv, hd, causal, sc, st, kl)
        torch.cuda.synchronize()
        fin = torch.isfinite(l1)
        c = {'case': name, 'finite': bool(torch.isfinite(o2.float()).all()), 'max_abs_dO': float((o1.float() - o2.float()).abs().max()),
             'rel_O': float((o1.float() - o2.float()).norm() / o1.float().norm()), 'max_abs_dlse': float((l1[fin] - l2[fin]).abs().max()),
             'lse_inf_pattern_equal': bool(torch.equal(fin, torch.isfinite(l2)))}
        c['ok'] = c['finite'] and c['rel_O'] < 5e-3 and c['max_abs_dO'] < 3e-2 and c['max_abs_dlse'] < 2e-3 and c['lse_inf_pattern_equal']
        if timed:
            fl = 4.0 * N * H * T * T * hd * (0.5 if causal else 1.0)
            for tag, fn in (('shipped', lambda: ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, causal, sc, start=st, kv_len=kl, out=o1)),
                            ('v2', lambda: fwd_v2(q, k, v, N, T, H, Hkv, hd, causal, sc, st, kl, out=o2))):
                for _ in range(3):
                    fn()
                best = 1e9
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / 20)
                c[tag + '_us'] = round(best * 1e3, 1)
                c[tag + '_tflops'] = round(fl / (best * 1e-3) / 1e12, 1)
        res['cases'].append(c)
        print(json.dumps(c), flush=True)
    print('ALL OK' if all(c['ok'] for c in res['cases']) else 'MISMATCH', flush=True)


if __name__ == '__main__':
    main()
