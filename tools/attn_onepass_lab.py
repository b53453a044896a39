"""Same-box A/B of the attention backward: the dQ + dK/dV kernel pair against the single-pass kernel (csrc/attn_bwd1.inc, aa_attn_bwd_onepass) at the
benchmarked geometry (causal 8 x 32 heads x 2048 x 128, with the rotary backward in the epilogues as the step runs it) and at Qwen2-VL's GQA 28 / 4.

    python tools/attn_onepass_lab.py [out.json]          timings (best of 3 x 20 launches) + max deviation between the two forms
    AA_LAB_ONLY=two|one python tools/attn_onepass_lab.py    under rocprofv3: only that form's launches of the bench block
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from align_anything_amd import ops
    dev = torch.device('cuda:0')
    only = os.environ.get('AA_LAB_ONLY')

    def case(N, T, H, Hkv, hd=128, seed=0, reps=20):
        g = torch.Generator(device='cpu').manual_seed(seed)
        W = (H + 2 * Hkv) * hd
        qkv = (torch.randn(N * T, W, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        q, k, v = qkv[:, :H * hd], qkv[:, H * hd:(H + Hkv) * hd], qkv[:, (H + Hkv) * hd:]
        do = (torch.randn(N * T, H * hd, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        sc = hd ** -0.5
        pos = torch.arange(T, dtype=torch.int32).repeat(N).to(dev)
        ang = torch.arange(T, dtype=torch.float32)[:, None] * (10000.0 ** (-torch.arange(0, hd, 2, dtype=torch.float32) / hd))[None, :]
        rope = (pos, ang.cos().to(torch.bfloat16).to(dev), ang.sin().to(torch.bfloat16).to(dev))
        o, lse = ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, True, sc)
        outs, res = {}, {}
        for name, flag in (('two', False), ('one', True)):
            if only and only != name:
                continue
            ops.ATTN_ONEPASS = flag
            d = torch.zeros_like(qkv)
            dq, dk, dv = d[:, :H * hd], d[:, H * hd:(H + Hkv) * hd], d[:, (H + Hkv) * hd:]
            fn = lambda: ops.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, N, T, H, Hkv, hd, True, sc, rope=rope)
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
            fl = 10.0 * N * H * T * T * hd * 0.5
            res[name + '_us'] = round(best * 1e3, 1)
            res[name + '_tflops'] = round(fl / best / 1e9, 1)
            outs[name] = d.float()
        if len(outs) == 2:
            a, b = outs['one'], outs['two']
            for nm, lo, hi_ in (('dq', 0, H * hd), ('dk', H * hd, (H + Hkv) * hd), ('dv', (H + Hkv) * hd, W)):
                res[nm + '_max_abs_diff'] = float((a[:, lo:hi_] - b[:, lo:hi_]).abs().max())
                res[nm + '_max_abs'] = float(b[:, lo:hi_].abs().max())
            res['finite'] = bool(torch.isfinite(a).all())
        return res

    out = {'bench_8x32x2048x128': case(8, 2048, 32, 32)}
    if not only:
        out['gqa_4x28of4x2048x128'] = case(4, 2048, 28, 4, seed=1)
        out['b1_2x32x2048x128'] = case(2, 2048, 32, 32, seed=2)
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', sys.argv[1]), 'w') as f:
            json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
