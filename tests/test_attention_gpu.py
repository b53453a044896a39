"""GPU: fused attention forward/backward (csrc/attention.hip) vs an explicit fp32 softmax reference."""
import math

import pytest
import torch

from tests.gpu_util import assert_close, dev, randn_bf16

pytestmark = pytest.mark.gpu


def ref_attention(q, k, v, do, N, T, H, Hkv, hd, causal, scale, start):
    """fp32 autograd reference on [N*T, H*hd] layouts; returns o, dq, dk, dv (pad rows zeroed)."""
    qf = q.float().view(N, T, H, hd).transpose(1, 2).detach().requires_grad_(True)
    kf = k.float().view(N, T, Hkv, hd).transpose(1, 2).detach().requires_grad_(True)
    vf = v.float().view(N, T, Hkv, hd).transpose(1, 2).detach().requires_grad_(True)
    rep = H // Hkv
    kk = kf.repeat_interleave(rep, 1); vv = vf.repeat_interleave(rep, 1)
    s = (qf @ kk.transpose(-1, -2)) * scale
    idx = torch.arange(T, device=q.device)
    mask = torch.zeros(N, 1, T, T, dtype=torch.bool, device=q.device)
    if causal:
        mask = mask | (idx[None, :] > idx[:, None])[None, None]
    valid = torch.ones(N, T, dtype=torch.bool, device=q.device)
    if start is not None:
        valid = idx[None, :] >= start[:, None].long()
        mask = mask | ~valid[:, None, None, :]
    s = s.masked_fill(mask, float('-inf'))
    row_ok = ~mask.all(-1, keepdim=True)
    p = torch.softmax(s.masked_fill(~row_ok, 0.0), -1) * row_ok
    o = p @ vv
    dof = do.float().view(N, T, H, hd).transpose(1, 2) * valid[:, None, :, None]
    (o * dof).sum().backward()
    back = lambda t, h: t.transpose(1, 2).reshape(N * T, h * hd)
    return back(o.detach(), H), back(qf.grad, H), back(kf.grad, Hkv), back(vf.grad, Hkv), valid.reshape(N * T)


CASES = [
    # N, T, H, Hkv, hd, causal, starts
    (2, 256, 2, 2, 128, True, [0, 37]),
    (1, 200, 2, 2, 128, True, [70]),
    (2, 577, 2, 2, 64, False, None),
    (2, 128, 4, 2, 64, True, [0, 5]),
    (1, 64, 1, 1, 128, True, None),
    (3, 48, 2, 2, 64, True, [0, 6, 2]),
]


@pytest.mark.parametrize('case', CASES)
def test_attention_forward_backward(case):
    from align_anything_amd import ops
    N, T, H, Hkv, hd, causal, starts = case
    scale = hd ** -0.5
    qkv = randn_bf16(N * T, (H + 2 * Hkv) * hd, seed=11)
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:(H + Hkv) * hd], qkv[:, (H + Hkv) * hd:]
    do = randn_bf16(N * T, H * hd, seed=12)
    start = torch.tensor(starts, dtype=torch.int32, device=dev()) if starts is not None else None
    o, lse = ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, causal, scale, start)
    dqkv = torch.zeros_like(qkv)
    dq, dk, dv = dqkv[:, :H * hd], dqkv[:, H * hd:(H + Hkv) * hd], dqkv[:, (H + Hkv) * hd:]
    # zero dO at pad rows, as the real backward does (no loss gradient reaches pad positions)
    ro, rdq, rdk, rdv, valid = ref_attention(q, k, v, do, N, T, H, Hkv, hd, causal, scale, start)
    do = do * valid[:, None].to(do.dtype)
    ops.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, N, T, H, Hkv, hd, causal, scale, start)
    torch.cuda.synchronize()
    assert torch.isfinite(o.float()).all()
    vm = valid[:, None].float()
    assert_close(o.float() * vm, ro * vm, rtol=2e-2, atol=2e-2, what=f'O {case}')
    assert float((o.float() * (1 - vm)).abs().max()) == 0.0, 'pad query rows must be exactly 0'
    g = max(float(rdq.abs().max()), 1e-3)
    assert_close(dq.float() * vm, rdq * vm, rtol=3e-2, atol=2e-2 * g, what=f'dQ {case}')
    assert_close(dk, rdk, rtol=3e-2, atol=2e-2 * max(float(rdk.abs().max()), 1e-3), what=f'dK {case}')
    assert_close(dv, rdv, rtol=3e-2, atol=2e-2 * max(float(rdv.abs().max()), 1e-3), what=f'dV {case}')
    # LSE against the reference on valid rows
    qf = q.float().view(N, T, H, hd).transpose(1, 2)
    kf = k.float().view(N, T, Hkv, hd).transpose(1, 2).repeat_interleave(H // Hkv, 1)
    s = (qf @ kf.transpose(-1, -2)) * scale
    idx = torch.arange(T, device=dev())
    m = torch.zeros(N, 1, T, T, dtype=torch.bool, device=dev())
    if causal:
        m = m | (idx[None, :] > idx[:, None])[None, None]
    if start is not None:
        m = m | ~(idx[None, :] >= start[:, None].long())[:, None, None, :]
    ref_lse = torch.logsumexp(s.masked_fill(m, float('-inf')), -1)
    ok = torch.isfinite(ref_lse)
    assert_close(lse[ok], ref_lse[ok], rtol=1e-3, atol=2e-2, what='lse')


@pytest.mark.parametrize('case', CASES)
def test_attention_backward_rope_epilogue_is_bit_identical(case):
    """aa_attn_bwd_rope (the rotary backward inside the dQ and dK/dV epilogues) == aa_attn_bwd followed by aa_rope_inplace(inverse=1) on
    dQ and dK, bit for bit: same bf16 rounding of dQ / dK before the rotation, same rounding points inside it (elementwise.hip::rope_kernel);
    dV untouched.  Positions are arbitrary per row (padding-left rows restart at 0 in the trainers)."""
    from align_anything_amd import ops
    N, T, H, Hkv, hd, causal, starts = case
    scale = hd ** -0.5
    qkv = randn_bf16(N * T, (H + 2 * Hkv) * hd, seed=21)
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:(H + Hkv) * hd], qkv[:, (H + Hkv) * hd:]
    do = randn_bf16(N * T, H * hd, seed=22)
    start = torch.tensor(starts, dtype=torch.int32, device=dev()) if starts is not None else None
    o, lse = ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, causal, scale, start)
    g = torch.Generator().manual_seed(5)
    pos = torch.randint(0, 4096, (N * T,), generator=g).to(torch.int32).to(dev())
    ang = torch.arange(4096, dtype=torch.float32)[:, None] * (10000.0 ** (-torch.arange(0, hd, 2, dtype=torch.float32) / hd))[None, :]
    cos_t, sin_t = ang.cos().to(torch.bfloat16).to(dev()), ang.sin().to(torch.bfloat16).to(dev())

    def run(fused):
        d = torch.zeros_like(qkv)
        dq, dk, dv = d[:, :H * hd], d[:, H * hd:(H + Hkv) * hd], d[:, (H + Hkv) * hd:]
        ops.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, N, T, H, Hkv, hd, causal, scale, start, rope=(pos, cos_t, sin_t) if fused else None)
        if not fused:
            ops.rope_(d, 0, H + Hkv, hd, pos, cos_t, sin_t, inverse=True)
        torch.cuda.synchronize()
        return d

    two, one = run(False), run(True)
    assert torch.isfinite(one.float()).all()
    assert float(one.float().abs().max()) > 0
    assert torch.equal(one.view(torch.int16), two.view(torch.int16)), f'{case}: max |d| {float((one.float() - two.float()).abs().max())}'
    with pytest.raises(RuntimeError):
        ops.attn_bwd(q, k, v, o, do, lse, one[:, :H * hd], one[:, H * hd:(H + Hkv) * hd], one[:, (H + Hkv) * hd:], N, T, H, Hkv, hd, causal, scale, start,
                     rope=(pos.long(), cos_t, sin_t))


@pytest.mark.parametrize('case', [(2, 1024, 4, 2, 128, [0, 100], [0, 700]), (3, 512, 2, 2, 128, None, [256, 0, 300]), (2, 384, 2, 2, 64, [0, 10], [0, 200])])
def test_attention_q_skip_leaves_out_only_what_nobody_consumes(case):
    """aa_attn_fwd_qskip / aa_attn_bwd_qskip (shared-prompt packing): query rows below q_skip[n] have no consumer and carry dO = 0.  Every row at or above
    q_skip[n] of O / lse, and ALL of dQ / dK / dV, equal the kernels without the skip (run on the same zeroed dO) bit for bit."""
    from align_anything_amd import ops
    N, T, H, Hkv, hd, starts, skips = case
    scale = hd ** -0.5
    qkv = randn_bf16(N * T, (H + 2 * Hkv) * hd, seed=41)
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:(H + Hkv) * hd], qkv[:, (H + Hkv) * hd:]
    do = randn_bf16(N * T, H * hd, seed=42)
    start = torch.tensor(starts, dtype=torch.int32, device=dev()) if starts is not None else None
    qs = torch.tensor(skips, dtype=torch.int32, device=dev())
    keep = (torch.arange(T, device=dev())[None, :] >= qs[:, None]).reshape(-1)
    do = do * keep[:, None].to(do.dtype)
    g = torch.Generator().manual_seed(5)
    pos = torch.randint(0, 4096, (N * T,), generator=g).to(torch.int32).to(dev())
    ang = torch.arange(4096, dtype=torch.float32)[:, None] * (10000.0 ** (-torch.arange(0, hd, 2, dtype=torch.float32) / hd))[None, :]
    rope = (pos, ang.cos().to(torch.bfloat16).to(dev()), ang.sin().to(torch.bfloat16).to(dev()))

    def run(skip, with_rope):
        o = torch.full((N * T, H * hd), float('nan'), dtype=torch.bfloat16, device=dev())
        o, lse = ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, True, scale, start, out=o, q_skip=qs if skip else None)
        oo = torch.where(keep[:, None], o, torch.zeros_like(o))          # what the consumer reads (the backward's delta = rowsum(dO O) sees O only where dO != 0)
        d = torch.full_like(qkv, float('nan'))
        ops.attn_bwd(q, k, v, oo, do, lse, d[:, :H * hd], d[:, H * hd:(H + Hkv) * hd], d[:, (H + Hkv) * hd:], N, T, H, Hkv, hd, True, scale, start,
                     rope=rope if with_rope else None, q_skip=qs if skip else None)
        torch.cuda.synchronize()
        return oo, lse, d

    for with_rope in (False, True):
        o0, lse0, d0 = run(False, with_rope)
        o1, lse1, d1 = run(True, with_rope)
        assert torch.equal(o0, o1)
        keep_h = keep.view(N, 1, T).expand(N, H, T)
        assert torch.equal(torch.nan_to_num(lse0[keep_h], neginf=-1e30), torch.nan_to_num(lse1[keep_h], neginf=-1e30))
        assert torch.isfinite(d1.float()).all()
        assert torch.equal(d0.float(), d1.float()), float((d0.float() - d1.float()).abs().max())          # (float compare: -0 == +0)
