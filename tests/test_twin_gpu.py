"""GPU: the bf16 PRODUCTION kernels against their fp32 parity-mode twins on IDENTICAL (bf16-representable) inputs.

The 1e-6 parity with the reference's fixtures is shown by the fp32 twins (gemm_f32.hip, attention_f32.hip, elementwise_f32.hip);
the bf16 kernels are what the benchmark times.  Against the fp32 fixtures they can only be held to the bf16 envelope (6e-2 on
gradients), under which a bf16-only bug could hide.  Here both families get the same inputs, so the only legitimate difference is
rounding: the bf16 result must be the fp32 result rounded to bf16, give or take the kernel's internal bf16 rounding points --
measured in bf16 ulps (2^-8 relative) of the larger of |value| and the row's RMS (cancellation makes tiny entries meaningless)."""
import pytest
import torch

from tests.gpu_util import dev, dump, randn_bf16

pytestmark = pytest.mark.gpu
ULP = 2.0 ** -8
REPORT = []


def ulps(got_bf16, want_f32, row_dim=-1):
    want = want_f32.float()
    # scale: the element, the RMS of its row, or (rows that are entirely ~0, e.g. masked query rows) the RMS of the whole tensor
    scale = torch.maximum(want.abs(), want.pow(2).mean(row_dim, keepdim=True).sqrt().expand_as(want))
    scale = torch.maximum(scale, want.pow(2).mean().sqrt() * 1e-2)
    return ((got_bf16.float() - want).abs() / (scale * ULP))


def check(name, got, want, max_ulp, frac_within2=0.99):
    u = ulps(got, want)
    mx, f2 = float(u.max()), float((u <= 2.0).float().mean())
    REPORT.append(f'{name}: max {mx:.2f} ulp, {100 * f2:.3f} % within 2 ulp')
    assert mx <= max_ulp and f2 >= frac_within2, REPORT[-1]


@pytest.mark.parametrize('layout', ['nt', 'nn', 'tn'])
def test_gemm_bf16_is_the_rounded_fp32_twin(layout):
    """Every 256-wide production tile kernel (8-wave gemm.hip, one-wave-per-SIMD gemm4.hip plain + general epilogue) vs aa_gemm_f32
    (exact fp32 MFMA): fp32 accumulation in both, so the bf16 output is within 1 ulp (summation order) of the rounded twin."""
    from align_anything_amd import ops
    a_t, b_n = layout == 'tn', layout in ('nn', 'tn')
    try:
        for (M, N, K) in [(512, 768, 512), (300, 264, 192), (1024, 512, 2048)]:
            if (a_t and M % 8) or (b_n and N % 8):
                continue
            a = randn_bf16(K, M, seed=1) if a_t else randn_bf16(M, K, seed=1)
            b = randn_bf16(K, N, seed=2) if b_n else randn_bf16(N, K, seed=2)
            want = ops.gemm(a.float(), b.float(), a_t=a_t, b_n=b_n, out_f32=True)
            for tile in (0, 5, 1):
                ops.gemm_set_tile(tile)
                check(f'gemm {layout} tile{tile} {M}x{N}x{K}', ops.gemm(a, b, a_t=a_t, b_n=b_n), want, max_ulp=1.01, frac_within2=1.0)
    finally:
        ops.gemm_set_tile(-1)


def test_block_kernels_bf16_vs_fp32_twin():
    """RMSNorm / LayerNorm / SwiGLU / RoPE, forward and backward: the second instantiation of the same source with elem_t = float."""
    from align_anything_amd import ops
    from align_anything_amd.modeling import rope_tables
    rows, h = 384, 1024
    x, w, dy = randn_bf16(rows, h, seed=1), (1 + 0.1 * torch.randn(h)).to(torch.bfloat16).to(dev()), randn_bf16(rows, h, seed=3)
    y16, r16 = ops.rmsnorm_fwd(x, w, 1e-5)
    y32, r32 = ops.rmsnorm_fwd(x.float(), w.float(), 1e-5)
    check('rmsnorm fwd', y16, y32, max_ulp=2.01)
    dw16, dw32 = torch.zeros(h, device=dev()), torch.zeros(h, device=dev())
    dx16 = ops.rmsnorm_bwd(dy, x, w, r16, dw16)
    dx32 = ops.rmsnorm_bwd(dy.float(), x.float(), w.float(), r32, dw32)
    check('rmsnorm bwd dx', dx16, dx32, max_ulp=2.01)
    assert float((dw16 - dw32).abs().max()) <= 2e-3 * float(dw32.abs().max())          # fp32 in both
    b = randn_bf16(h, seed=4)
    l16, m16, s16 = ops.layernorm_fwd(x, w, b, 1e-5)
    l32, m32, s32 = ops.layernorm_fwd(x.float(), w.float(), b.float(), 1e-5)
    check('layernorm fwd', l16, l32, max_ulp=2.01)
    gu, dact = randn_bf16(rows, 2 * h, seed=5), randn_bf16(rows, h, seed=6)
    check('swiglu fwd', ops.swiglu_fwd(gu), ops.swiglu_fwd(gu.float()), max_ulp=2.01)
    check('swiglu bwd', ops.swiglu_bwd(gu, dact), ops.swiglu_bwd(gu.float(), dact.float()), max_ulp=2.01)
    H, hd = 4, 128
    q = randn_bf16(rows, H * hd, seed=7)
    pos = (torch.arange(rows, dtype=torch.int32, device=dev()) * 3) % 200
    c16, s16_ = rope_tables(256, hd, 10000.0, dev())
    c32, s32_ = rope_tables(256, hd, 10000.0, dev(), torch.float32)
    q16, q32 = q.clone(), q.float()
    ops.rope_(q16, 0, H, hd, pos, c16, s16_)
    ops.rope_(q32, 0, H, hd, pos, c32.to(torch.bfloat16).float(), s32_.to(torch.bfloat16).float())   # identical (bf16-valued) tables
    check('rope', q16, q32, max_ulp=3.01)                               # three bf16 rounding points (HF's) vs one


def test_attention_bf16_vs_fp32_twin():
    """Flash attention forward / backward, hd 128 and 64, causal with left padding: P is rounded to bf16 before P V in the production
    kernel (like HF's `attn_weights.to(v.dtype)`), so individual outputs may sit a few ulps from the twin; the bulk must be within 2."""
    from align_anything_amd import ops
    for (N, T, H, Hkv, hd, starts) in [(2, 512, 4, 4, 128, [0, 77]), (2, 320, 4, 2, 64, [5, 0])]:
        scale = hd ** -0.5
        qkv = randn_bf16(N * T, (H + 2 * Hkv) * hd, seed=11, scale=0.8)
        q, k, v = qkv[:, :H * hd], qkv[:, H * hd:(H + Hkv) * hd], qkv[:, (H + Hkv) * hd:]
        do = randn_bf16(N * T, H * hd, seed=12)
        start = torch.tensor(starts, dtype=torch.int32, device=dev())
        valid = (torch.arange(T, device=dev())[None, :] >= start[:, None].long()).reshape(N * T, 1)
        do = do * valid.to(do.dtype)
        outs = []
        for dt in (torch.bfloat16, torch.float32):
            qq, kk, vv, dd = (t.to(dt).contiguous() for t in (q, k, v, do))
            o, lse = ops.attn_fwd(qq, kk, vv, N, T, H, Hkv, hd, True, scale, start)
            dq, dk, dv = (torch.zeros_like(t) for t in (qq, kk, vv))
            ops.attn_bwd(qq, kk, vv, o, dd, lse, dq, dk, dv, N, T, H, Hkv, hd, True, scale, start)
            vm = valid.to(dt)
            outs.append([o * vm, dq * vm, dk, dv, lse])            # pad query rows carry don't-care values
        for name, a, b, mx in zip(('O', 'dQ', 'dK', 'dV'), outs[0], outs[1], (8.0, 16.0, 16.0, 16.0)):
            check(f'attention hd{hd} {name}', a, b, max_ulp=mx, frac_within2=0.97)
        ok = torch.isfinite(outs[1][4])
        assert float((outs[0][4][ok] - outs[1][4][ok]).abs().max()) < 2e-2


def test_zz_write_twin_report():
    dump('parity_bf16_vs_fp32_twin_ulps.txt', '\n'.join(REPORT) + '\n')
    assert len(REPORT) >= 10
