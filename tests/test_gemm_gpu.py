"""GPU: bf16 MFMA GEMM (csrc/gemm.hip) vs a plain fp32 torch matmul of the same bf16 operands."""
import pytest
import torch

from tests.gpu_util import assert_close, dev, randn_bf16

pytestmark = pytest.mark.gpu

SHAPES = [(128, 128, 64), (256, 256, 128), (300, 200, 192), (1154, 1024, 640), (1024, 2048, 1024), (64, 320, 128)]


def _ref(a, b, a_t, b_n):
    A = a.float().t() if a_t else a.float()
    B = b.float() if b_n else b.float().t()
    return A @ B


@pytest.mark.parametrize('tile', [0, 1, 2, 3, 5])
@pytest.mark.parametrize('layout', ['nt', 'nn', 'tn'])
def test_gemm_layouts_and_tiles(tile, layout):
    """tile 0-3: 16x16x32-MFMA tile configs of the 8-wave kernel; 5: the one-wave-per-SIMD kernel
    with accumulator-file MFMAs (gemm4.hip).  (Its 32x32x16-MFMA sibling passed these tests in round 3 and lost the A/B: profiles/r03_gemm5_mfma32_negative.txt; source in git history, commit b2c2a50.)"""
    from align_anything_amd import ops
    ops.gemm_set_tile(tile)
    try:
        for (M, N, K) in SHAPES:
            a_t = layout == 'tn'
            b_n = layout in ('nn', 'tn')
            if a_t and M % 8:
                continue
            if b_n and N % 8:
                continue
            a = randn_bf16(K, M, seed=1) if a_t else randn_bf16(M, K, seed=1)
            b = randn_bf16(K, N, seed=2) if b_n else randn_bf16(N, K, seed=2)
            out = ops.gemm(a, b, a_t=a_t, b_n=b_n)
            torch.cuda.synchronize()
            ref = _ref(a, b, a_t, b_n)
            assert_close(out, ref, rtol=1e-2, atol=1e-2 * float(ref.abs().mean()), what=f'{layout} tile{tile} {M}x{N}x{K}')
    finally:
        ops.gemm_set_tile(-1)


@pytest.mark.parametrize('mfma32', [False, 'g4'])
def test_gemm_epilogues_match_hf_rounding_points(mfma32):
    """'g4' = the general epilogue of the one-wave-per-SIMD kernel (gemm4.hip; its plain bf16 epilogue is what the layout test runs)."""
    from align_anything_amd import ops
    g4 = mfma32 == 'g4'
    ops.gemm_set_tile(5 if g4 else -1)
    M, N, K = 384, 512, 256
    a, w = randn_bf16(M, K, seed=3), randn_bf16(N, K, scale=0.1, seed=4)
    bias, res = randn_bf16(N, seed=5), randn_bf16(M, N, seed=6)
    acc = a.float() @ w.float().t()
    # bias + quick_gelu (CLIP fc1): bf16(act(bf16(acc + bias)))
    out = ops.gemm(a, w, bias=bias, act=ops.ACT_QUICK_GELU)
    y = (acc + bias.float()).to(torch.bfloat16).float()
    ref = (y * torch.sigmoid(1.702 * y))
    assert_close(out, ref, rtol=1e-2, atol=2e-2, what='bias+quick_gelu')
    # bias + erf gelu
    out = ops.gemm(a, w, bias=bias, act=ops.ACT_GELU)
    assert_close(out, torch.nn.functional.gelu(y), rtol=1e-2, atol=2e-2, what='bias+gelu')
    # residual: bf16(bf16(acc) + res)
    out = ops.gemm(a, w, residual=res)
    assert_close(out, acc.to(torch.bfloat16).float() + res.float(), rtol=1e-2, atol=2e-2, what='residual')
    # in-place residual (C aliases residual)
    buf = res.clone()
    ops.gemm(a, w, out=buf, residual=buf)
    assert_close(buf, acc.to(torch.bfloat16).float() + res.float(), rtol=1e-2, atol=2e-2, what='residual inplace')
    # accumulate + fp32 out
    c32 = torch.ones((M, N), dtype=torch.float32, device=dev())
    ops.gemm(a, w, out=c32, accumulate=True)
    assert_close(c32, acc + 1.0, rtol=1e-3, atol=1e-2, what='accum f32')
    cb = res.clone()
    ops.gemm(a, w, out=cb, accumulate=True)
    assert_close(cb, acc + res.float(), rtol=1e-2, atol=3e-2, what='accum bf16')
    # strided views (column slices of a wider buffer), as the fused qkv / gate_up buffers are used
    wide = randn_bf16(M, 2 * K, seed=7)
    out = ops.gemm(wide[:, K:], w)
    assert_close(out, wide[:, K:].float() @ w.float().t(), rtol=1e-2, atol=2e-2, what='strided A')
    ops.gemm_set_tile(-1)


def test_gemm_rejects_bad_arguments_loudly():
    from align_anything_amd import ops
    from align_anything_amd.lib import AAHipError
    a, b = randn_bf16(64, 100), randn_bf16(64, 100)
    with pytest.raises(AAHipError):
        ops.gemm(a, b)  # K not a multiple of 64


@pytest.mark.parametrize('layout', ['nt', 'nn', 'tn'])
def test_gemm_8wave_256_tile_on_ragged_and_short_k_shapes(layout):
    """The 8-wave 256x256 kernel (the fallback of gemm4 when K is not a multiple of 128, and the grouped-GEMM kernel) on ragged and short-K
    shapes (1, 2, 3 K-tiles) with the schedule each layout ships with (NN: peeled full interleave, NT / TN: phase-A interleave; the other
    schedules of rounds 1-2 are no longer instantiated)."""
    from align_anything_amd import ops
    ops.gemm_set_tile(0)
    try:
        for (M, N, K) in [(264, 520, 64), (256, 256, 128), (300, 200, 192), (1154, 1024, 640), (512, 768, 2048),
                          (520, 264, 256), (256, 512, 320), (2048, 1024, 4096)]:
            a_t, b_n = layout == 'tn', layout in ('nn', 'tn')
            if (a_t and M % 8) or (b_n and N % 8):
                continue
            a = randn_bf16(K, M, seed=1) if a_t else randn_bf16(M, K, seed=1)
            b = randn_bf16(K, N, seed=2) if b_n else randn_bf16(N, K, seed=2)
            out = ops.gemm(a, b, a_t=a_t, b_n=b_n)
            ref = _ref(a, b, a_t, b_n)
            assert_close(out, ref, rtol=1e-2, atol=1e-2 * float(ref.abs().mean()), what=f'{layout} {M}x{N}x{K}')
    finally:
        ops.gemm_set_tile(-1)


def test_gemm4_ring_pipeline_and_fast_epilogues():
    """gemm4.hip's specialised paths: the 4-slot LDS ring over short and long contractions (1, 2, 3, 5, 7 trips
    of four stages; more tiles than CUs), plain and residual epilogues with 16-byte permlane-swapped stores; K = 192 is not a multiple of
    the ring's trip and takes the 8-wave kernel of the same tile."""
    from align_anything_amd import ops
    ops.gemm_set_tile(5)
    try:
        for (M, N, K) in [(512, 512, 256), (256, 256, 128), (256, 256, 384), (4096, 8192, 640), (16384, 4096, 384), (8192, 4352, 896), (512, 768, 192)]:
            for layout in ('nt', 'nn', 'tn'):
                a_t, b_n = layout == 'tn', layout in ('nn', 'tn')
                a = randn_bf16(K, M, seed=1) if a_t else randn_bf16(M, K, seed=1)
                b = randn_bf16(K, N, seed=2) if b_n else randn_bf16(N, K, seed=2)
                out = ops.gemm(a, b, a_t=a_t, b_n=b_n)
                rows = torch.arange(0, M, 61, device=a.device)
                ref = ((a[:, rows].t() if a_t else a[rows]).float()) @ (b if b_n else b.t()).float()
                assert_close(out[rows], ref, rtol=1e-2, atol=1e-2 * float(ref.abs().mean()), what=f'{layout} {M}x{N}x{K}')
                cs = out.double().sum(0)
                want = (a.double().sum(1) if a_t else a.double().sum(0)) @ (b if b_n else b.t()).double()
                assert float((cs - want).abs().max()) < 0.04 * (M ** 0.5) * float(ref.abs().mean()) * 8 + 1e-2 * float(want.abs().mean()), (layout, M, N, K)
            # residual epilogue (forward layout): bf16(bf16(acc) + res), also in place
            a, w, res = randn_bf16(M, K, seed=3), randn_bf16(N, K, scale=0.1, seed=4), randn_bf16(M, N, seed=6)
            acc = a.float() @ w.float().t()
            want = (acc.to(torch.bfloat16).float() + res.float()).to(torch.bfloat16)
            out = ops.gemm(a, w, residual=res)
            assert_close(out, want, rtol=1e-2, atol=7e-2, what=f'residual {M}x{N}x{K}')     # one bf16 ulp of the accumulator where acc + res cancels
            frac_exact = float((out == want).float().mean())
            assert frac_exact > 0.995, frac_exact          # same rounding points: only fp32 summation order differs from the torch matmul
            buf = res.clone()
            ops.gemm(a, w, out=buf, residual=buf)
            assert torch.equal(buf, out)
    finally:
        ops.gemm_set_tile(-1)


def test_fused_epilogues_are_bit_identical_to_the_unfused_kernels():
    """aa_gemm_qkv_rope_bf16 / aa_gemm_glu_fwd_bf16 / aa_gemm_glu_bwd_bf16: the GEMM with HF's rotary embedding / SwiGLU forward /
    SwiGLU backward in its epilogue (gemm4.hip) against the same entry point with fusion switched off (GEMM kernel + aa_rope_inplace /
    aa_swiglu_fwd / aa_swiglu_bwd): same rounding points, so EVERY bit must agree; shapes that do not qualify take the unfused path."""
    from align_anything_amd import ops
    from align_anything_amd.modeling import rope_tables
    g = torch.Generator(device='cpu').manual_seed(5)
    try:
        for (M, K, H, Hkv) in [(512, 256, 4, 2), (256, 128, 2, 2), (1024, 640, 3, 1), (256, 64, 2, 2), (320, 128, 2, 1)]:       # last two: K % 128 / M % 256 != 0 -> unfused inside
            hd = 128
            N = (H + 2 * Hkv) * hd
            x, w = randn_bf16(M, K, seed=1), randn_bf16(N, K, scale=0.2, seed=2)
            pos = torch.randint(0, 300, (M,), generator=g).to(torch.int32).to(dev())
            cos_t, sin_t = rope_tables(320, hd, 10000.0, dev())
            outs = []
            for fuse in (True, False):
                ops.gemm_set_fuse(fuse)
                outs.append(ops.gemm_qkv_rope(x, w, pos, cos_t, sin_t, H + Hkv, hd))
            assert torch.equal(outs[0], outs[1]), ('rope', M, K, H, Hkv, float((outs[0].float() - outs[1].float()).abs().max()))
            # and against the definition: rotate_half form on the bf16 projection
            ops.gemm_set_fuse(True)
            y = (x.float() @ w.float().t()).to(torch.bfloat16)
            qk = y[:, :(H + Hkv) * hd].float().view(M, H + Hkv, hd)
            c = torch.cat([cos_t[pos.long()], cos_t[pos.long()]], -1).float()[:, None]
            s_ = torch.cat([sin_t[pos.long()], sin_t[pos.long()]], -1).float()[:, None]
            rot = torch.cat([-qk[..., hd // 2:], qk[..., :hd // 2]], -1)
            ref = (qk * c).to(torch.bfloat16).float() + (rot * s_).to(torch.bfloat16).float()
            assert_close(outs[0][:, :(H + Hkv) * hd].float().view(M, H + Hkv, hd), ref, rtol=1.6e-2, atol=7e-2, what='rope vs definition')       # one bf16 ulp of the projection where the two terms cancel
            assert_close(outs[0][:, (H + Hkv) * hd:], y[:, (H + Hkv) * hd:], rtol=1e-2, atol=2e-2, what='v heads untouched')
        for (M, K, F) in [(512, 256, 384), (256, 128, 128), (768, 384, 1408), (768, 320, 1408), (320, 128, 256)]:
            x, w = randn_bf16(M, K, seed=3), randn_bf16(2 * F, K, scale=0.2, seed=4)
            res = []
            for fuse in (True, False):
                ops.gemm_set_fuse(fuse)
                res.append(ops.gemm_glu_fwd(x, w, F))
            assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), ('glu fwd', M, K, F)
            gu = (x.float() @ w.float().t()).to(torch.bfloat16)
            assert_close(res[0][0], gu, rtol=1e-2, atol=2e-2, what='gate|up')
            want = (torch.nn.functional.silu(gu[:, :F].float()).to(torch.bfloat16).float() * gu[:, F:].float())
            assert_close(res[0][1], want, rtol=1.6e-2, atol=2e-2, what='silu(gate)*up')
        for (M, K, F) in [(512, 256, 512), (256, 64, 256), (1024, 192, 768), (1024, 384, 768), (320, 128, 256), (512, 128, 384)]:
            dy, wd = randn_bf16(M, K, seed=6), randn_bf16(K, F, scale=0.2, seed=7)
            gu = randn_bf16(M, 2 * F, seed=8)
            res = []
            for fuse in (True, False):
                ops.gemm_set_fuse(fuse)
                res.append(ops.gemm_glu_bwd(dy, wd, gu, F))
            assert torch.equal(res[0], res[1]), ('glu bwd', M, K, F, float((res[0].float() - res[1].float()).abs().max()))
            dact = (dy.float() @ wd.float()).to(torch.bfloat16)
            assert_close(res[0], ops.swiglu_bwd(gu, dact), rtol=1.6e-2, atol=2e-2, what='glu bwd vs unfused pieces')
    finally:
        ops.gemm_set_fuse(True)


@pytest.mark.parametrize('case', ['nt_residual', 'nt_bias_act', 'nn', 'tn_f32_accumulate'])
def test_split_k_for_few_row_launches_matches_the_one_launch_kernel(case, monkeypatch):
    """ops.gemm cuts the contraction of a few-row launch ([~800, K] against a 7B weight matrix: a PPO rollout's scoring forwards, ppo.py:224-289) into chunks that
    run side by side (aa_gemm_splitk_bf16) and sums their fp32 partial products in order.  Against the one-launch kernel (AA_GEMM_SPLITK off): equal up to the
    fp32 association of the accumulator -- at most one bf16 ulp on an output --, and both within the usual bound of the fp32 reference."""
    from align_anything_amd import ops
    from tests.util import rel_err
    g = torch.Generator(device='cpu').manual_seed(11)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).to(dev())
    M = 832
    kw, out0 = {}, None
    if case == 'nt_residual':           # down projection of Qwen2-VL-7B at one 832-token sequence
        a, b = mk(M, 18944), mk(3584, 18944); kw = dict(residual=mk(M, 3584))
    elif case == 'nt_bias_act':
        a, b = mk(M, 3584), mk(4608, 3584); kw = dict(bias=mk(4608), act=1)
    elif case == 'nn':                  # gate_up input gradient: K = 37888
        a, b = mk(M, 37888), mk(37888, 3584); kw = dict(b_n=True)
    else:                               # a weight gradient with few output rows, fp32 accumulate
        a, b = mk(4096, 1024), mk(4096, 3584); kw = dict(a_t=True, b_n=True, accumulate=True)
        out0 = torch.randn(1024, 3584, generator=g).to(dev())
    got = []
    for on in (False, True):
        monkeypatch.setattr(ops, 'SPLITK', on)
        Mq, Nq, Kq = (a.shape[1] if kw.get('a_t') else a.shape[0]), (b.shape[1] if kw.get('b_n') else b.shape[0]), (a.shape[0] if kw.get('a_t') else a.shape[1])
        assert bool(ops._splitk_chunks(Mq, Nq, Kq)) == on
        got.append(ops.gemm(a, b, out=out0.clone() if out0 is not None else None, **kw).float())
    torch.cuda.synchronize()
    ref = (a.float().t() if kw.get('a_t') else a.float()) @ (b.float() if kw.get('b_n') else b.float().t())
    if 'bias' in kw:
        ref = ref + kw['bias'].float()
    if kw.get('act') == 1:
        ref = torch.nn.functional.gelu(ref.to(torch.bfloat16).float())
    if 'residual' in kw:
        ref = ref.to(torch.bfloat16).float() + kw['residual'].float()
    if out0 is not None:
        ref = ref + out0
    e = rel_err(got[1], got[0])
    assert e < (1e-6 if out0 is not None else 2e-3), e
    assert rel_err(got[1], ref) < 6e-3 and rel_err(got[0], ref) < 6e-3
