"""Micro-benchmarks of the hot kernels on one MI355X (run on the GPU box via gpurun).

Writes gpurun_out/bench_kernels.json: achieved TFLOP/s (GEMM, attention) and GB/s (HBM-bound kernels)
against the gfx950 peaks from /opt/skills/guides/MI355X_MICROARCH.md (2.5 PFLOP/s dense bf16, 8 TB/s).
torch.matmul (hipBLASLt) is timed beside the hand-written GEMM as an external yardstick only.
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from align_anything_amd import ops  # noqa: E402

PEAK_TF, PEAK_GBS = 2500.0, 8000.0
dev = torch.device('cuda:0')


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms


def rnd(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).to(torch.bfloat16)


def bench_gemm(res, quick):
    shapes = [('qkv', 4096, 12288, 4096), ('o', 4096, 4096, 4096), ('gate_up', 4096, 22016, 4096),
              ('down', 4096, 4096, 11008), ('lm_head', 1024, 32064, 4096), ('sq8k', 8192, 8192, 8192)]
    if quick:
        shapes = shapes[:2]
    for name, M, N, K in shapes:
        flops = 2.0 * M * N * K
        for layout in ('nt', 'nn', 'tn'):
            a_t, b_n = layout == 'tn', layout in ('nn', 'tn')
            a = rnd(K, M) if a_t else rnd(M, K)
            b = rnd(K, N) if b_n else rnd(N, K)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            for tile in (0, 1, 2, 3):
                ops.gemm_set_tile(tile)
                try:
                    ms = timeit(lambda: ops.gemm(a, b, out=out, a_t=a_t, b_n=b_n))
                    res.append(dict(kernel='gemm', name=name, layout=layout, tile=tile, M=M, N=N, K=K, ms=ms,
                                    tflops=flops / ms / 1e9, frac_peak=flops / ms / 1e9 / PEAK_TF))
                except Exception as ex:  # noqa
                    res.append(dict(kernel='gemm', name=name, layout=layout, tile=tile, error=str(ex)))
            ops.gemm_set_tile(-1)
            ms = timeit(lambda: ops.gemm(a, b, out=out, a_t=a_t, b_n=b_n))
            res.append(dict(kernel='gemm', name=name, layout=layout, tile='auto', M=M, N=N, K=K, ms=ms,
                            tflops=flops / ms / 1e9, frac_peak=flops / ms / 1e9 / PEAK_TF))
            A = a.t() if a_t else a
            B = b if b_n else b.t()
            ms = timeit(lambda: torch.matmul(A, B, out=out))
            res.append(dict(kernel='torch.matmul', name=name, layout=layout, M=M, N=N, K=K, ms=ms,
                            tflops=flops / ms / 1e9))
            print(res[-2], flush=True); print(res[-1], flush=True)


def bench_attn(res):
    for (N, T, H, hd, causal) in [(2, 2048, 32, 128, True), (4, 577, 16, 64, False)]:
        qkv = rnd(N * T, 3 * H * hd)
        q, k, v = qkv[:, :H * hd], qkv[:, H * hd:2 * H * hd], qkv[:, 2 * H * hd:]
        o, lse = ops.attn_fwd(q, k, v, N, T, H, H, hd, causal, hd ** -0.5)
        f = 4.0 * N * H * T * T * hd * (0.5 if causal else 1.0)
        ms = timeit(lambda: ops.attn_fwd(q, k, v, N, T, H, H, hd, causal, hd ** -0.5, out=o))
        res.append(dict(kernel='attn_fwd', N=N, T=T, H=H, hd=hd, causal=causal, ms=ms, tflops=f / ms / 1e9))
        do = rnd(N * T, H * hd); dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[:, :H * hd], dqkv[:, H * hd:2 * H * hd], dqkv[:, 2 * H * hd:]
        ms = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, N, T, H, H, hd, causal, hd ** -0.5))
        res.append(dict(kernel='attn_bwd', N=N, T=T, H=H, hd=hd, causal=causal, ms=ms, tflops=2.5 * f / ms / 1e9))
        print(res[-2], flush=True); print(res[-1], flush=True)


def bench_mem(res):
    M, h, F, V = 4096, 4096, 11008, 32064
    x, w = rnd(M, h), rnd(h)
    y, rstd = ops.rmsnorm_fwd(x, w, 1e-5)
    ms = timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-5, out=y, rstd=rstd))
    res.append(dict(kernel='rmsnorm_fwd', ms=ms, gbs=2 * M * h * 2 / ms / 1e6))
    dw = torch.zeros(h, device=dev); dx = torch.empty_like(x)
    ms = timeit(lambda: ops.rmsnorm_bwd(y, x, w, rstd, dw, dx=dx))
    res.append(dict(kernel='rmsnorm_bwd', ms=ms, gbs=3 * M * h * 2 / ms / 1e6))
    gu = rnd(M, 2 * F); act = ops.swiglu_fwd(gu)
    ms = timeit(lambda: ops.swiglu_fwd(gu, out=act))
    res.append(dict(kernel='swiglu_fwd', ms=ms, gbs=3 * M * F * 2 / ms / 1e6))
    dgu = torch.empty_like(gu)
    ms = timeit(lambda: ops.swiglu_bwd(gu, act, out=dgu))
    res.append(dict(kernel='swiglu_bwd', ms=ms, gbs=5 * M * F * 2 / ms / 1e6))
    rows = 1022
    logits = rnd(rows, V); labels = torch.randint(0, V, (rows,), device=dev)
    lp, lse = ops.logprob_gather_fwd(logits, labels)
    ms = timeit(lambda: ops.logprob_gather_fwd(logits, labels))
    res.append(dict(kernel='logprob_fwd', rows=rows, V=V, ms=ms, gbs=rows * V * 2 / ms / 1e6))
    d = torch.empty_like(logits)
    ms = timeit(lambda: ops.logprob_gather_bwd(logits, labels, lse, lp, out=d))
    res.append(dict(kernel='logprob_bwd', rows=rows, V=V, ms=ms, gbs=2 * rows * V * 2 / ms / 1e6))
    n = 512 * 1024 * 1024
    master = torch.zeros(n, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    p16 = torch.zeros(n, dtype=torch.bfloat16, device=dev); g = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    ms = timeit(lambda: ops.adamw_flat_(master, m, v, p16, g, 1e-6, 0.9, 0.95, 1e-8, 0.0, 1), iters=5)
    res.append(dict(kernel='adamw_flat', n=n, ms=ms, gbs=28.0 * n / ms / 1e6))
    xt = torch.empty(h, M, dtype=torch.bfloat16, device=dev)
    ms = timeit(lambda: ops.transpose(x, out=xt))
    res.append(dict(kernel='transpose', ms=ms, gbs=2 * M * h * 2 / ms / 1e6))
    for r in res[-9:]:
        r['frac_hbm_peak'] = r['gbs'] / PEAK_GBS
        print(r, flush=True)


if __name__ == '__main__':
    quick = '--quick' in sys.argv
    res = []
    t0 = time.time()
    for fn in (bench_mem, bench_attn):
        try:
            fn(res)
        except Exception as ex:  # keep going: one failing kernel must not hide the others
            res.append(dict(kernel=fn.__name__, error=repr(ex)))
            print('ERROR', fn.__name__, ex, flush=True)
    try:
        if '--no-gemm' not in sys.argv:
            bench_gemm(res, quick)
    except Exception as ex:
        res.append(dict(kernel='gemm', error=repr(ex)))
        print('ERROR gemm', ex, flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'bench_kernels.json'), 'w') as f:
        json.dump(res, f, indent=1)
    print('done in', time.time() - t0, 's')
