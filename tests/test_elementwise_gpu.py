"""GPU: HBM-bound block kernels (csrc/elementwise.hip) vs plain fp32 torch references."""
import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import assert_close, dev, randn_bf16

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('rows,h', [(37, 128), (300, 4096), (64, 1024), (5, 11008 - 11008 % 8), (1000, 128), (37, 64), (4099, 512)])
def test_rmsnorm_fwd_bwd(rows, h):
    from align_anything_amd import ops
    x, w, dy = randn_bf16(rows, h, seed=1), randn_bf16(h, seed=2) * 0.5 + 1, randn_bf16(rows, h, seed=3)
    y, rstd = ops.rmsnorm_fwd(x, w, 1e-5)
    xf = x.float().requires_grad_(True); wf = w.float().requires_grad_(True)
    xn = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)
    ref = wf * xn
    assert_close(y, ref, rtol=1.6e-2, atol=1e-3, what='rmsnorm y')
    ref.backward(dy.float())
    dw = torch.zeros(h, dtype=torch.float32, device=dev())
    dx = ops.rmsnorm_bwd(dy, x, w, rstd, dw)
    torch.cuda.synchronize()
    assert_close(dx, xf.grad, rtol=2e-2, atol=2e-2 * float(xf.grad.abs().max()), what='rmsnorm dx')
    assert_close(dw, wf.grad, rtol=2e-2, atol=2e-2 * float(wf.grad.abs().max()), what='rmsnorm dw')
    # add_to_dx accumulates into the residual-stream gradient
    base = randn_bf16(rows, h, seed=4)
    acc = base.clone()
    ops.rmsnorm_bwd(dy, x, w, rstd, None, dx=acc, add_to_dx=True)
    assert_close(acc, xf.grad + base.float(), rtol=2e-2, atol=3e-2 * float(xf.grad.abs().max() + 1), what='rmsnorm dx+=')


@pytest.mark.parametrize('rows,h', [(37, 128), (200, 768), (64, 1024)])
def test_layernorm_fwd_bwd(rows, h):
    from align_anything_amd import ops
    x, dy = randn_bf16(rows, h, seed=1) * 2 + 0.5, randn_bf16(rows, h, seed=3)
    w, b = randn_bf16(h, seed=2) * 0.5 + 1, randn_bf16(h, seed=5)
    y, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-5)
    xf, wf, bf = x.float().requires_grad_(True), w.float().requires_grad_(True), b.float().requires_grad_(True)
    ref = F.layer_norm(xf, (h,), wf, bf, 1e-5)
    assert_close(y, ref, rtol=1.6e-2, atol=1e-2, what='layernorm y')
    ref.backward(dy.float())
    dw = torch.zeros(h, dtype=torch.float32, device=dev()); db = torch.zeros_like(dw)
    dx = ops.layernorm_bwd(dy, x, w, mean, rstd, dw, db)
    torch.cuda.synchronize()
    assert_close(dx, xf.grad, rtol=2e-2, atol=2e-2 * float(xf.grad.abs().max()), what='ln dx')
    assert_close(dw, wf.grad, rtol=2e-2, atol=2e-2 * float(wf.grad.abs().max()), what='ln dw')
    assert_close(db, bf.grad, rtol=2e-2, atol=2e-2 * float(bf.grad.abs().max()), what='ln db')


def test_rope_matches_hf_formula_and_inverse_is_transpose():
    from align_anything_amd import ops
    from oracle import models as om
    N, T, H, hd = 2, 40, 3, 128
    buf = randn_bf16(N * T, 3 * H * hd, seed=9)
    orig = buf.clone()
    cos, sin = om.rope_tables(T, hd, 10000.0)
    cos_b, sin_b = cos.to(torch.bfloat16).to(dev()), sin.to(torch.bfloat16).to(dev())
    pos = torch.arange(T, dtype=torch.int32, device=dev()).repeat(N)
    ops.rope_(buf, 0, H, hd, pos, cos_b, sin_b)          # q heads
    ops.rope_(buf, H * hd, H, hd, pos, cos_b, sin_b)      # k heads
    torch.cuda.synchronize()
    q = orig[:, :H * hd].float().view(N, T, H, hd).transpose(1, 2)
    ref = om.apply_rope(q, cos_b.float(), sin_b.float()).transpose(1, 2).reshape(N * T, H * hd)
    assert_close(buf[:, :H * hd], ref, rtol=1e-2, atol=1e-2, what='rope q')
    assert torch.equal(buf[:, 2 * H * hd:], orig[:, 2 * H * hd:]), 'v columns must be untouched'
    # inverse(forward(x)) ~= x (rotation), and inverse is the exact autograd transpose
    g = randn_bf16(N * T, H * hd, seed=10)
    gi = g.clone()
    ops.rope_(gi, 0, H, hd, pos, cos_b, sin_b, inverse=True)
    qq = q.clone().requires_grad_(True)
    om.apply_rope(qq, cos_b.float(), sin_b.float()).backward(g.float().view(N, T, H, hd).transpose(1, 2))
    assert_close(gi, qq.grad.transpose(1, 2).reshape(N * T, H * hd), rtol=1e-2, atol=1e-2, what='rope bwd')


def test_swiglu_and_pointwise_activations():
    from align_anything_amd import ops
    M, Fd = 130, 1376
    gu, dact = randn_bf16(M, 2 * Fd, seed=1), randn_bf16(M, Fd, seed=2)
    out = ops.swiglu_fwd(gu)
    g = gu[:, :Fd].float().requires_grad_(True); u = gu[:, Fd:].float().requires_grad_(True)
    ref = F.silu(g) * u
    assert_close(out, ref, rtol=1.6e-2, atol=1e-2, what='swiglu')
    ref.backward(dact.float())
    d = ops.swiglu_bwd(gu, dact)
    assert_close(d[:, :Fd], g.grad, rtol=2e-2, atol=2e-2, what='swiglu dgate')
    assert_close(d[:, Fd:], u.grad, rtol=2e-2, atol=2e-2, what='swiglu dup')
    x, dy = randn_bf16(64, 512, seed=3), randn_bf16(64, 512, seed=4)
    for code, fn in ((ops.ACT_GELU, F.gelu), (ops.ACT_QUICK_GELU, lambda t: t * torch.sigmoid(1.702 * t)),
                     (ops.ACT_RELU, F.relu), (ops.ACT_SILU, F.silu)):
        xf = x.float().requires_grad_(True)
        r = fn(xf)
        assert_close(ops.act_fwd(x, code), r, rtol=1e-2, atol=1e-2, what=f'act{code}')
        r.backward(dy.float())
        assert_close(ops.act_bwd(x, dy, code), xf.grad, rtol=2e-2, atol=2e-2, what=f'dact{code}')
    a, b = randn_bf16(33, 256, seed=5), randn_bf16(33, 256, seed=6)
    assert torch.equal(ops.add(a, b), (a.float() + b.float()).to(torch.bfloat16))


def test_embedding_gather_and_image_scatter_are_bit_exact():
    """hf:models/llava/modeling_llava.py:234-248 semantics; pure indexing -> bit-exact."""
    from align_anything_amd import ops
    V, h, N, T, IMG = 320, 128, 3, 50, 300
    E, feat = randn_bf16(V, h, seed=1), randn_bf16(17, h, seed=2)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 299, (N, T), generator=g)
    # 17 image tokens scattered irregularly over the rows
    flat = ids.view(-1)
    where = torch.randperm(N * T, generator=g)[:17]
    flat[where] = IMG
    ids = ids.to(dev())
    slot, count = ops.image_slot_index(ids.view(-1), IMG)
    x = ops.embed_fwd(ids.view(-1), E, slot, feat)
    torch.cuda.synchronize()
    assert int(count.item()) == 17
    ref = F.embedding(ids, E)
    mask = (ids == IMG)
    ref = ref.masked_scatter(mask[..., None].expand_as(ref), feat)
    assert torch.equal(x.view(N, T, h), ref)
    exp_slot = torch.where(mask.view(-1), torch.cumsum(mask.view(-1).int(), 0) - 1, torch.full_like(mask.view(-1).int(), -1))
    assert torch.equal(slot, exp_slot.int())
    # backward: dfeat = gather of dx rows (exact), dE = scatter-add of the rest
    dx = randn_bf16(N * T, h, seed=3)
    dE = torch.zeros(V, h, dtype=torch.float32, device=dev())
    dfeat = torch.zeros(17, h, dtype=torch.bfloat16, device=dev())
    ops.embed_bwd(ids.view(-1), dx, V, slot=slot, dE=dE, dfeat=dfeat)
    torch.cuda.synchronize()
    assert torch.equal(dfeat, dx[mask.view(-1)])
    refE = torch.zeros(V, h, dtype=torch.float32, device=dev())
    refE.index_add_(0, ids.view(-1)[~mask.view(-1)], dx.float()[~mask.view(-1)])
    assert_close(dE, refE, rtol=1e-5, atol=1e-5, what='dE')
    # empty image set and learned positions (OPT)
    pos = torch.randint(0, 60, (N * T,), generator=g).int().to(dev())
    P = randn_bf16(64, h, seed=7)
    ids2 = torch.randint(0, V, (N * T,), generator=g).to(dev())
    y = ops.embed_fwd(ids2, E, pos=pos, P=P)
    assert torch.equal(y, (E[ids2].float() + P[pos.long()].float()).to(torch.bfloat16))


def test_transpose_colsum_im2col_clip_embed():
    from align_anything_amd import ops
    x = randn_bf16(1022, 320, seed=1)
    xt = ops.transpose(x, pad_cols_to=1024)
    assert torch.equal(xt[:, :1022], x.t()) and float(xt[:, 1022:].abs().max()) == 0.0
    x2 = randn_bf16(70, 130, seed=2)
    assert torch.equal(ops.transpose(x2), x2.t().contiguous())
    cs = torch.zeros(320, dtype=torch.float32, device=dev())
    ops.colsum_(x, cs)
    assert_close(cs, x.float().sum(0), rtol=1e-4, atol=1e-3, what='colsum')
    # CLIP patch conv == im2col + GEMM
    n_img, P, Himg, hdim = 3, 14, 28, 128
    pix = torch.randn(n_img, 3, Himg, Himg, generator=torch.Generator().manual_seed(3)).to(dev())
    w = randn_bf16(hdim, 3, P, P, scale=0.05, seed=4)
    Kp = 640
    col = ops.patch_im2col(pix, P, Kp)
    wp = torch.zeros(hdim, Kp, dtype=torch.bfloat16, device=dev()); wp[:, :588] = w.view(hdim, -1)
    pe = ops.gemm(col, wp)
    ref = F.conv2d(pix.to(torch.bfloat16).float(), w.float(), stride=P).flatten(2).transpose(1, 2).reshape(-1, hdim)
    assert_close(pe, ref, rtol=1e-2, atol=1e-2, what='patch embed')
    cls, pos = randn_bf16(hdim, seed=5), randn_bf16(5, hdim, seed=6)
    emb = ops.clip_embed(pe, cls, pos, n_img, 4)
    ref2 = torch.cat([cls.float().expand(n_img, 1, hdim), pe.float().view(n_img, 4, hdim)], 1) + pos.float()[None]
    assert_close(emb.view(n_img, 5, hdim), ref2, rtol=1e-2, atol=1e-2, what='clip embed')


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('hd,heads,tokens', [(128, 32, 200), (128, 4, 777), (64, 6, 130)])
def test_rmsnorm_rope_fused_is_bit_identical_to_the_pair(hd, heads, tokens, dtype):
    """aa_rmsnorm_rope_fwd (Qwen3's per-head q_norm / k_norm + apply_rotary_pos_emb in one pass) == aa_rmsnorm_fwd followed by aa_rope_inplace, bit for bit,
    in bf16 (fp32 twin: to fma-contraction rounding); ragged row counts (the last block's partial row pairs)."""
    from align_anything_amd import ops
    from align_anything_amd.modeling import rope_tables
    g = torch.Generator().manual_seed(hd + heads)
    x = (torch.randn(tokens * heads, hd, generator=g) * 2.0).to(dtype).to(dev())
    w = (1.0 + 0.1 * torch.randn(hd, generator=g)).to(dtype).to(dev())
    pos = torch.randint(0, 300, (tokens,), generator=g).to(torch.int32).to(dev())
    cos, sin = rope_tables(300, hd, 10000.0, dev(), dtype)
    y0, r0 = ops.rmsnorm_fwd(x, w, 1e-6)
    y0 = y0.view(tokens, heads * hd).clone()
    ops.rope_(y0, 0, heads, hd, pos, cos, sin)
    y1, r1 = ops.rmsnorm_rope_fwd(x, w, 1e-6, pos, cos, sin, heads)
    torch.cuda.synchronize()
    assert torch.equal(r0, r1)
    if dtype == torch.bfloat16:
        assert torch.equal(y0.view(-1, hd), y1)
    else:       # the fp32 twin has no rounding points between the multiplies and the add: the compiler's fma contraction may differ between the two kernels
        assert torch.allclose(y0.view(-1, hd), y1, rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_per_head_norm_on_column_slices_of_the_fused_projection(dtype):
    """The Qwen3-MoE stack projects q | k | v with ONE GEMM; the per-head norm (+ rotary embedding) reads its column slice of that output in place and its
    backward writes dq / dk into their slices of the fused gradient (aa_rmsnorm_rope_fwd with ldx, aa_rmsnorm_heads_bwd): same bits as the dense kernels on
    contiguous copies, nothing written outside the slice."""
    from align_anything_amd import ops
    from align_anything_amd.modeling import rope_tables
    tokens, H, Hkv, hd = 333, 8, 2, 128
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(tokens, (H + 2 * Hkv) * hd, generator=g) * 1.5).to(dtype).to(dev())
    pos = torch.randint(0, 200, (tokens,), generator=g).to(torch.int32).to(dev())
    cos, sin = rope_tables(200, hd, 10000.0, dev(), dtype)
    for heads, lo in ((H, 0), (Hkv, H * hd)):
        w = (1.0 + 0.1 * torch.randn(hd, generator=g)).to(dtype).to(dev())
        sl = qkv[:, lo:lo + heads * hd]
        dense = sl.contiguous().view(tokens * heads, hd)
        y0, r0 = ops.rmsnorm_rope_fwd(dense, w, 1e-6, pos, cos, sin, heads)
        y1, r1 = ops.rmsnorm_rope_fwd(sl, w, 1e-6, pos, cos, sin, heads, hd=hd)
        assert torch.equal(y0, y1) and torch.equal(r0, r1)
        dy = (torch.randn(tokens * heads, hd, generator=g) * 0.3).to(dtype).to(dev())
        dw0 = torch.zeros(hd, dtype=torch.float32, device=dev()); dw1 = torch.zeros_like(dw0)
        dx0 = ops.rmsnorm_bwd(dy, dense, w, r0, dw0)
        d_qkv = torch.full_like(qkv, 7.0)
        ops.rmsnorm_heads_bwd(dy, sl, w, r1, dw1, d_qkv[:, lo:lo + heads * hd], heads, hd)
        torch.cuda.synchronize()
        assert torch.equal(d_qkv[:, lo:lo + heads * hd].contiguous().view(-1, hd), dx0) and torch.allclose(dw0, dw1, rtol=1e-5, atol=1e-5)      # dw: fp32 atomics across blocks
        keep = torch.ones(qkv.shape[1], dtype=torch.bool); keep[lo:lo + heads * hd] = False
        assert bool((d_qkv[:, keep.to(dev())] == 7.0).all())
