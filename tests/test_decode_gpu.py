"""GPU: decode kernels (csrc/decode.hip) and the native rollout (align_anything_amd/generation.py) that replaces
HF generate in the PPO loop (align_anything/trainers/text_to_text/ppo.py:209-222)."""
import pytest
import torch

from oracle import models as om
from tests.gpu_util import assert_close, dev, dump, randn_bf16
from tests.util import load_golden, state_dict_from_golden, tiny_llava_cfg, tiny_opt_cfg

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.mark.parametrize('M', [1, 3, 16])
def test_skinny_gemm(M):
    from align_anything_amd import ops
    for (N, K) in [(64, 128), (320, 640), (4112, 4096), (1000, 11008)]:
        x, w = randn_bf16(M, K, seed=1), randn_bf16(N, K, scale=0.1, seed=2)
        bias, res = randn_bf16(N, seed=3), randn_bf16(M, N, seed=4)
        acc = x.float() @ w.float().t()
        assert_close(ops.linear_small(x, w), acc, rtol=1e-2, atol=1e-2 * float(acc.abs().mean()) + 1e-3, what=f'skinny {M}x{N}x{K}')
        out = ops.linear_small(x, w, bias=bias, residual=res)
        ref = (acc + bias.float()).to(torch.bfloat16).float() + res.float()
        assert_close(out, ref, rtol=1e-2, atol=2e-2, what='skinny bias+residual')


@pytest.mark.parametrize('M', [1, 5, 16])
def test_skinny_gemm_on_strip_major_weights_is_bit_identical(M):
    """aa_swizzle_weights_bf16 + aa_gemm_skinny_swz_bf16 (the rollout's weight layout): same operands, same MFMA order."""
    from align_anything_amd import ops
    for (N, K) in [(64, 128), (320, 640), (1000, 2048), (12288, 4096), (4096, 11008), (32064, 4096)]:
        x, w = randn_bf16(M, K, seed=1), randn_bf16(N, K, scale=0.1, seed=2)
        bias, res = randn_bf16(N, seed=3), randn_bf16(M, N, seed=4)
        sw = ops.SwizzledWeight(w)
        assert sw.shape == (N, K) and sw.data.numel() == (N + 15) // 16 * 16 * K
        # the layout itself (integer work): element (n, k) sits at ((n/16)*(K/32) + k/32)*512 + ((n%16) + 16*((k%32)/8))*8 + k%8
        n_i = torch.tensor([0, 1, 15, 16, N - 1, N // 2]); k_i = torch.tensor([0, 7, 8, 31, 32, K - 1])
        off = ((n_i // 16) * (K // 32) + k_i // 32) * 512 + ((n_i % 16) + 16 * ((k_i % 32) // 8)) * 8 + k_i % 8
        assert torch.equal(sw.data.cpu()[off], w.cpu()[n_i, k_i])
        assert torch.equal(ops.linear_small(x, sw), ops.linear_small(x, w)), (M, N, K)
        assert torch.equal(ops.linear_small(x, sw, bias=bias, residual=res), ops.linear_small(x, w, bias=bias, residual=res)), (M, N, K)
    with pytest.raises(RuntimeError):
        ops.linear_small(randn_bf16(17, 128, seed=1), ops.SwizzledWeight(randn_bf16(64, 128, seed=2)))


@pytest.mark.parametrize('M', [1, 5, 16])
def test_skinny_gemm_with_folded_rmsnorm_and_swiglu(M):
    """aa_gemm_skinny_fused_bf16: the decode step's RMSNorm / SwiGLU ride in the weight stream that consumes them."""
    from align_anything_amd import ops
    ops.DECODE_FUSED = True          # the folded variant is off by default (slower at 7B); exercised here
    try:
        _check_folded(ops, M)
    finally:
        ops.DECODE_FUSED = False


def _check_folded(ops, M):
    for (N, K) in [(64, 128), (320, 640), (12288, 4096), (4096, 11008), (1000, 2048)]:
        x, w = randn_bf16(M, K, scale=1.5, seed=1), randn_bf16(N, K, scale=0.05, seed=2)
        nw = (1 + 0.2 * torch.randn(K, generator=torch.Generator().manual_seed(3))).to(torch.bfloat16).to(dev())
        bias, res = randn_bf16(N, seed=4), randn_bf16(M, N, seed=5)
        # RMSNorm prologue vs fp64 math of norm -> matmul; the unfused native chain is the second yardstick
        xn = x.double() * torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + 1e-5) * nw.double()
        ref = xn @ w.double().t()
        got = ops.linear_small(x, w, norm=(nw, 1e-5))
        unf = ops.linear_small(ops.rmsnorm_fwd(x, nw, 1e-5)[0], w)
        tol = 1e-2 * float(ref.abs().mean()) + 1e-3
        assert_close(got, ref.float(), rtol=1e-2, atol=tol, what=f'rmsnorm->skinny {M}x{N}x{K}')
        assert float((got.double() - ref).abs().mean()) < 1.5 * float((unf.double() - ref).abs().mean()) + 1e-4      # no worse than the unfused chain
        got = ops.linear_small(x, w, bias=bias, residual=res, norm=(nw, 1e-5))
        assert_close(got, (ref.float() + bias.float()).to(torch.bfloat16).float() + res.float(), rtol=1e-2, atol=tol + 2e-2, what='rmsnorm->skinny bias+residual')
        # SwiGLU prologue: the fragments are exactly aa_swiglu_fwd's values and the accumulation order is the same -> bit-equal
        gu = randn_bf16(M, 2 * K, seed=6)
        got = ops.linear_small(gu, w, residual=res, swiglu=True)
        assert torch.equal(got, ops.linear_small(ops.swiglu_fwd(gu), w, residual=res)), (M, N, K)


def test_decode_rope_cache_equals_rope_then_index_put():
    from align_anything_amd import ops
    from align_anything_amd.modeling import rope_tables
    for (N, H, Hkv, hd, Tmax) in [(3, 4, 2, 64, 50), (16, 32, 32, 128, 40), (5, 28, 4, 128, 33)]:
        kw = Hkv * hd
        qkv = randn_bf16(N, (H + 2 * Hkv) * hd, seed=1)
        cos, sin = rope_tables(64, hd, 10000.0, dev(), torch.bfloat16)
        pos = torch.randint(0, 64, (N,), generator=torch.Generator().manual_seed(2)).to(torch.int32).to(dev())
        slot = torch.randint(0, Tmax, (N,), generator=torch.Generator().manual_seed(3)).to(dev())
        a, ca = qkv.clone(), randn_bf16(N * Tmax, 2 * kw, seed=4)
        b, cb = qkv.clone(), ca.clone()
        ops.decode_rope_cache(a, H, Hkv, hd, pos, cos, sin, ca, Tmax, slot)
        ops.rope_(b, 0, H + Hkv, hd, pos, cos, sin)
        cb.view(N, Tmax, 2 * kw).index_put_((torch.arange(N, device=dev()), slot), b[:, H * hd:])
        assert torch.equal(a[:, :H * hd], b[:, :H * hd]) and torch.equal(ca, cb), (N, H, Hkv, hd)


@pytest.mark.parametrize('hd,H,Hkv,N,Tmax', [(128, 4, 4, 3, 300), (64, 4, 2, 3, 300), (128, 28, 4, 1, 900), (128, 32, 8, 5, 300), (64, 32, 32, 16, 300)])
def test_decode_attention_vs_reference(hd, H, Hkv, N, Tmax):
    """aa_attn_decode against softmax(q K^T) V in fp32 on every launch form: H N < 128 (one or two sequences: four key steps in flight per wave,
    round 6, aa_decode_set_rules), < 512 (8 waves, two steps; four under rule bit 2), beyond (4 waves); left-padded and full rows, a one-key row."""
    from align_anything_amd import ops
    q = randn_bf16(N, H * hd, seed=1)
    cache = randn_bf16(N * Tmax, 2 * Hkv * hd, seed=2)
    kw = Hkv * hd
    s_list = [0, 17, Tmax - 180, 3, 0][:N] + [5 * i % 40 for i in range(max(0, N - 5))]
    l_list = [Tmax, 131, Tmax - 179, Tmax - 1, 64][:N] + [Tmax - 7 * i % 50 for i in range(max(0, N - 5))]
    if N == 1:
        s_list, l_list = [0], [Tmax - 70]
    start = torch.tensor(s_list, dtype=torch.int32, device=dev())
    length = torch.tensor(l_list, dtype=torch.int32, device=dev())
    outs = {}
    old = ops.decode_set_rules(0)
    try:
        for mask in (0, 1, 5):          # two key steps in flight; four when H N < 128; four when H N < 512
            ops.decode_set_rules(mask)
            outs[mask] = ops.attn_decode(q, cache, cache[:, kw:], Tmax, start, length, N, H, Hkv, hd, hd ** -0.5)
    finally:
        ops.decode_set_rules(old)
    torch.cuda.synchronize()
    if H * N >= 128:
        assert torch.equal(outs[0], outs[1])
    if H * N >= 512:
        assert torch.equal(outs[0], outs[5])
    cf = cache.float().view(N, Tmax, 2, Hkv, hd)
    for n in range(N):
        s0, s1 = int(start[n]), int(length[n])
        assert 0 <= s0 < s1 <= Tmax
        for h in range(H):
            hk = h // (H // Hkv)
            k = cf[n, s0:s1, 0, hk]; v = cf[n, s0:s1, 1, hk]
            p = torch.softmax((k @ q[n, h * hd:(h + 1) * hd].float()) * hd ** -0.5, 0)
            for mask, o in outs.items():
                assert_close(o[n, h * hd:(h + 1) * hd], p @ v, rtol=2e-2, atol=1e-2, what=f'decode attn n{n} h{h} rules {mask}')


def test_token_selection_kernels():
    from align_anything_amd import ops
    V = 32064
    logits = randn_bf16(6, V, scale=2.0, seed=5)
    logits[0, 77] = 30.0; logits[0, 12345] = 30.0          # tie -> first index
    logits[1, V - 1] = 40.0
    idx = ops.argmax_rows(logits)
    assert idx.tolist() == torch.argmax(logits.float(), dim=-1).tolist() and idx[0].item() == 77
    # repetition penalty: seen tokens get score < 0 ? score * p : score / p before the argmax
    bitmap = torch.zeros(6, V, dtype=torch.uint8, device=dev())
    seen_ids = torch.randint(0, V, (6, 4000), generator=torch.Generator().manual_seed(3)).to(dev())
    seen_ids[0, 0] = 77; seen_ids[1, 0] = V - 1; seen_ids[2, 1] = -5; seen_ids[3, 2] = V + 9     # out-of-range ids are ignored
    ops.mark_seen_(bitmap, seen_ids)
    want_map = torch.zeros(6, V, dtype=torch.bool)
    for r in range(6):
        ok = seen_ids[r].cpu(); ok = ok[(ok >= 0) & (ok < V)]
        want_map[r, ok] = True
    assert torch.equal(bitmap.cpu().bool(), want_map)
    lf = logits.float().cpu()
    pen_ref = torch.where(want_map, torch.where(lf < 0, lf * 1.5, lf / 1.5), lf)
    idx = ops.argmax_rows(logits, bitmap, 1.5)
    assert idx.tolist() == torch.argmax(pen_ref, dim=-1).tolist() and idx[0].item() == 12345   # 77 is penalised, its twin wins
    # nucleus sampling against a straightforward CPU implementation of the same rule (HF TopPLogitsWarper:
    # smallest set of top tokens with mass >= top_p; draw by inverse CDF in index order with the given u)
    small = randn_bf16(64, 500, scale=3.0, seed=6)
    u = torch.rand(64, generator=torch.Generator().manual_seed(1)).to(dev())
    seen = (torch.rand(64, 500, generator=torch.Generator().manual_seed(2)) < 0.3).to(torch.uint8).to(dev())
    for temp, top_p, pen in ((1.0, 1.0, 1.0), (0.7, 0.9, 1.0), (1.3, 0.5, 1.0), (0.9, 0.8, 1.7), (1.0, 1.0, 0.6)):
        got = ops.sample_top_p(small, temp, top_p, u, seen if pen != 1.0 else None, pen).cpu()
        eff = small.float().cpu()
        if pen != 1.0:   # hf RepetitionPenaltyLogitsProcessor on the fp32 scores, before temperature / top-p
            eff = torch.where(seen.cpu().bool(), torch.where(eff < 0, eff * pen, eff / pen), eff)
        p = torch.softmax(eff / temp, -1)
        for r in range(64):
            sp, si = torch.sort(p[r], descending=True)
            k = int((torch.cumsum(sp, 0) < top_p).sum()) + 1 if top_p < 1.0 else p.shape[1]
            keep = torch.zeros_like(p[r], dtype=torch.bool); keep[si[:k]] = True
            pk = torch.where(keep, p[r], torch.zeros_like(p[r]))
            c = torch.cumsum(pk, 0)
            want = int((c > u[r].item() * pk.sum()).nonzero()[0])
            if got[r].item() != want:   # tolerate only a draw that lands on a boundary of the CDF / the kept set
                assert keep[got[r]] or abs(float(p[r][got[r]] - sp[k - 1])) < 1e-6, (temp, top_p, r)
                assert abs(float(c[got[r]] - c[want])) < 1e-3 * float(pk.sum()) + float(p[r][got[r]]) + float(p[r][want])


def test_top_k_cut_matches_hf_warpers_in_hf_order():
    """TopKLogitsWarper between temperature and top-p (hf:generation/logits_process.py; the reference's GenerationConfig inherits HF's default
    top_k = 50 under transformers 4.x, trainers/text_to_text/ppo.py:161-170): the kept set must be EXACTLY HF's -- every score >= the k-th
    largest, ties kept -- and the draw the inverse CDF of the renormalised rest.  Checked through HF's own warper classes on the CPU."""
    from transformers.generation.logits_process import (RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper,
                                                        TopPLogitsWarper)
    from align_anything_amd import ops
    rows, V = 48, 3001
    g = torch.Generator().manual_seed(11)
    logits = (torch.randn(rows, V, generator=g) * 4.0).to(torch.bfloat16)
    logits[3, 5:60] = logits[3].float().max().to(torch.bfloat16)            # 55 tied maxima: k = 50 keeps all 55 (HF removes only strictly smaller scores)
    logits[4] = logits[4].abs() * -1.0                                      # all-negative row (the key order of negative floats)
    lg = logits.to(dev())
    u = torch.rand(rows, generator=g).to(dev())
    seen = (torch.rand(rows, V, generator=g) < 0.2)
    ids = [seen[r].nonzero().reshape(1, -1) for r in range(rows)]
    for temp, top_k, top_p, pen in ((1.0, 50, 1.0, 1.0), (0.7, 50, 0.9, 1.0), (1.2, 1, 1.0, 1.0), (0.9, 7, 0.6, 1.3), (1.0, 3000, 0.95, 1.0), (1.0, 0, 0.9, 1.0)):
        got = ops.sample_top_p(lg, temp, top_p, u, seen.to(torch.uint8).to(dev()) if pen != 1.0 else None, pen, top_k=top_k).cpu()
        for r in range(rows):
            sc = logits[r:r + 1].float()
            if pen != 1.0:
                sc = RepetitionPenaltyLogitsProcessor(pen)(ids[r], sc)
            sc = TemperatureLogitsWarper(temp)(None, sc)
            if top_k:
                sc = TopKLogitsWarper(top_k)(None, sc)
            kept_k = torch.isfinite(sc[0])
            if top_p < 1.0:
                sc = TopPLogitsWarper(top_p)(None, sc)
            p = torch.softmax(sc[0], -1)
            c = torch.cumsum(p, 0)
            want = int((c > u[r].item()).nonzero()[0]) if bool((c > u[r].item()).any()) else int(p.nonzero()[-1])
            assert bool(kept_k[got[r]]), (temp, top_k, top_p, r, 'drew a token the top-k cut removes')
            if got[r].item() != want:      # only a draw on a boundary of the CDF / of the nucleus may differ (fp32 summation order)
                assert abs(float(c[got[r]] - c[want])) < 2e-3 + float(p[got[r]]) + float(p[want]) or float(p[got[r]]) > 0, (temp, top_k, top_p, r)
        if top_k == 1:
            assert torch.equal(got[:3], logits[:3].float().argmax(-1)) and 5 <= int(got[3]) < 60      # k = 1 -> greedy (a tie: any of the tied maxima)
    # exactness of the cut on its own (top_p = 1: everything the cut keeps can be drawn, nothing else): sweep u over the CDF of row 3
    us = torch.linspace(0.0, 0.99999, 400).to(dev())
    row = lg[3:4].expand(400, V).contiguous()
    drawn = set(ops.sample_top_p(row, 1.0, 1.0, us, None, 1.0, top_k=50).cpu().tolist())
    assert drawn == set((logits[3].float() == logits[3].float().max()).nonzero().reshape(-1).tolist()) and len(drawn) >= 55      # every tied maximum (the 55 set above + the original one), nothing else


def _greedy_oracle(logits_fn, ids, mask, n_new, eos, pad, penalty=1.0):
    ids, mask = ids.clone(), mask.clone()
    unfinished = torch.ones(ids.shape[0], dtype=torch.bool)
    margins = []
    for _ in range(n_new):
        lg = logits_fn(ids, mask)[:, -1]
        if penalty != 1.0:   # hf RepetitionPenaltyLogitsProcessor.__call__
            sc = torch.gather(lg, 1, ids)
            lg = lg.scatter(1, ids, torch.where(sc < 0, sc * penalty, sc / penalty))
        top2 = torch.topk(lg, 2, dim=-1).values
        margins.append(top2[:, 0] - top2[:, 1])
        nxt = torch.where(unfinished, lg.argmax(-1), torch.full((ids.shape[0],), pad))
        ids = torch.cat([ids, nxt[:, None]], 1)
        mask = torch.cat([mask, torch.ones(ids.shape[0], 1, dtype=mask.dtype)], 1)
        if eos is not None:
            unfinished = unfinished & (nxt != eos)
    return ids, torch.stack(margins, 1)


def _check_greedy(model, logits_fn, ids, mask, n_new, eos, pad, tag, pixel_values=None, penalty=1.0):
    from align_anything_amd.generation import generate
    seq = generate(model, ids.to(dev()), mask.to(dev()), max_new_tokens=n_new, do_sample=False, eos_token_id=eos,
                   pad_token_id=pad, pixel_values=pixel_values, sync_every=2, repetition_penalty=penalty).cpu()
    want, margins = _greedy_oracle(logits_fn, ids, mask, n_new, eos, pad, penalty)
    Tn = ids.shape[1]
    assert torch.equal(seq[:, :Tn], ids)
    agree = 0
    for n in range(ids.shape[0]):
        for s in range(seq.shape[1] - Tn):
            if seq[n, Tn + s] == want[n, Tn + s]:
                agree += 1
                continue
            # bf16 decode vs fp32 oracle may flip a near-tie; anything else is a bug
            assert margins[n, s] < 0.08, (tag, n, s, float(margins[n, s]), int(seq[n, Tn + s]), int(want[n, Tn + s]))
            break
    dump(f'parity_generate_{tag}.txt', f'native {seq.tolist()}\noracle {want.tolist()}\nagree {agree}\n')
    assert agree >= ids.shape[0] * 3
    return seq


def test_generate_greedy_opt_and_eos_padding():
    from align_anything_amd.modeling import build_model
    z = load_golden('opt_tiny_dpo.npz')
    cfg = tiny_opt_cfg()
    m = build_model(cfg, 'cuda:0', trainable=False)
    m.load_state_dict(state_dict_from_golden(z, 'w.', torch.bfloat16))
    sd = state_dict_from_golden(z, 'w.')
    ids, mask = T(z['input_ids'])[:, :24].clone(), T(z['attention_mask'])[:, :24].clone()
    fn = lambda i, a: om.opt_logits(sd, cfg, i, a)
    seq = _check_greedy(m, fn, ids, mask, 12, None, 1, 'opt')
    assert seq.shape == (4, 36)
    # EOS: pick the token row 0 emits at step 2 as EOS -> row 0 must be padded afterwards
    eos = int(seq[0, 24 + 2])
    seq2 = _check_greedy(m, fn, ids, mask, 12, eos, 1, 'opt_eos')
    first = (seq2[0, 24:] == eos).nonzero()[0].item()
    assert (seq2[0, 24 + first + 1:] == 1).all()
    # pad_token_id == eos_token_id (Llama / Qwen tokenizers): HF stops after the step in which the last row emitted EOS and KEEPS
    # that column -- the length is counted in steps, not inferred from non-pad columns
    from align_anything_amd.generation import generate
    one = generate(m, ids[:1].to(dev()), mask[:1].to(dev()), max_new_tokens=12, do_sample=False, eos_token_id=eos, pad_token_id=eos,
                   sync_every=1).cpu()
    assert one.shape == (1, 24 + first + 1) and int(one[0, -1]) == eos and torch.equal(one[0], seq2[0, :24 + first + 1])
    late = generate(m, ids[:1].to(dev()), mask[:1].to(dev()), max_new_tokens=12, do_sample=False, eos_token_id=eos, pad_token_id=eos,
                    sync_every=5).cpu()          # the loop notices late; the trim is the same
    assert torch.equal(late, one)
    # repetition penalty (hf RepetitionPenaltyLogitsProcessor): a strong penalty forbids repeats and changes the path
    seq3 = _check_greedy(m, fn, ids, mask, 12, None, 1, 'opt_reppen', penalty=4.0)
    assert not torch.equal(seq3, seq)


def test_generate_greedy_llava_with_image_prefill():
    from align_anything_amd.modeling import build_model
    z = load_golden('llava_tiny_dpo.npz')
    cfg = tiny_llava_cfg()
    m = build_model(cfg, 'cuda:0', trainable=False)
    m.load_state_dict(state_dict_from_golden(z, 'w.', torch.bfloat16))
    sd = state_dict_from_golden(z, 'w.')
    ids, mask, pix = T(z['input_ids'])[:, :30].clone(), T(z['attention_mask'])[:, :30].clone(), T(z['pixel_values'])
    fn = lambda i, a: om.llava_logits(sd, cfg, i, a, pix)
    import os
    _check_greedy(m, fn, ids, mask, 8, None, 301, 'llava', pixel_values=pix.to(dev()))          # default: strip-major copies with the norms folded in
    assert m.stack.prepare_decode(4)[0]['qkv'].folded and m.head.prepare_decode(4).folded
    os.environ['AA_DECODE_NORM_FOLD'] = '0'
    try:
        seq = _check_greedy(m, fn, ids, mask, 8, None, 301, 'llava_unfolded', pixel_values=pix.to(dev()))
        # the rollout ran on strip-major weight copies (LlamaStack.prepare_decode); the row-major kernels give the same tokens
        assert m.stack.prepare_decode(4) is not None and not m.stack.prepare_decode(4)[0]['qkv'].folded
        os.environ['AA_DECODE_SWIZZLE'] = '0'
        try:
            assert m.stack.prepare_decode(4) is None
            assert torch.equal(_check_greedy(m, fn, ids, mask, 8, None, 301, 'llava_rowmajor', pixel_values=pix.to(dev())), seq)
        finally:
            del os.environ['AA_DECODE_SWIZZLE']
    finally:
        del os.environ['AA_DECODE_NORM_FOLD']


def test_generate_sampling_runs_and_respects_length_cap():
    from align_anything_amd.generation import generate
    from align_anything_amd.modeling import build_model
    z = load_golden('opt_tiny_dpo.npz')
    m = build_model(tiny_opt_cfg(), 'cuda:0', trainable=False)
    m.load_state_dict(state_dict_from_golden(z, 'w.', torch.bfloat16))
    ids, mask = T(z['input_ids'])[:, :20].to(dev()), T(z['attention_mask'])[:, :20].to(dev())
    g = torch.Generator(device='cuda').manual_seed(0)
    a = generate(m, ids, mask, max_length=32, do_sample=True, temperature=0.8, top_p=0.9, pad_token_id=1, generator=g)
    g = torch.Generator(device='cuda').manual_seed(0)
    b = generate(m, ids, mask, max_length=32, do_sample=True, temperature=0.8, top_p=0.9, pad_token_id=1, generator=g)
    assert a.shape == (4, 32) and torch.equal(a, b) and int(a.max()) < 320 and int(a.min()) >= 0
    g = torch.Generator(device='cuda').manual_seed(0)
    c = generate(m, ids, mask, max_length=32, do_sample=True, temperature=0.8, top_p=0.9, pad_token_id=1, generator=g, repetition_penalty=1.3)
    assert c.shape == (4, 32) and torch.equal(c[:, :20], ids) and not torch.equal(c, a)


def test_generate_replaying_a_hipgraph_gives_the_same_tokens():
    """generate(use_graph=True): one decode position captured once and replayed (state lives in static device buffers)."""
    from align_anything_amd.generation import generate
    from align_anything_amd.modeling import build_model
    for fixture, cfg, pad, pix_key in (('opt_tiny_dpo.npz', tiny_opt_cfg(), 1, None), ('llava_tiny_dpo.npz', tiny_llava_cfg(), 301, 'pixel_values')):
        z = load_golden(fixture)
        m = build_model(cfg, 'cuda:0', trainable=False)
        m.load_state_dict(state_dict_from_golden(z, 'w.', torch.bfloat16))
        n = 24 if pix_key is None else 30
        ids, mask = T(z['input_ids'])[:, :n].to(dev()), T(z['attention_mask'])[:, :n].to(dev())
        kw = dict(max_new_tokens=10, pad_token_id=pad, pixel_values=T(z[pix_key]).to(dev()) if pix_key else None)
        for sample in (False, True):
            outs = []
            for ug in (False, True):
                g = torch.Generator(device='cuda').manual_seed(7)
                outs.append(generate(m, ids, mask, do_sample=sample, temperature=0.9, top_p=0.8, generator=g, use_graph=ug, **kw))
                assert generate.last_used_graph == ug, (fixture, sample, ug)
            assert torch.equal(outs[0], outs[1]), (fixture, sample)


@pytest.mark.parametrize('M', [1, 4, 16])
def test_strip_kernel_fused_epilogues_are_bit_identical_to_the_kernel_pairs(M):
    """Round 3: the rollout's strip kernel finishes the element-wise kernel that follows the projection in its epilogue, made possible by
    permuting the rows inside every strip of the rollout-only weight copy (aa_swizzle_weights_perm_bf16):
      * [gate; up] weight, mode 'glu': aa_gemm_skinny_swz_glu_bf16 == aa_gemm_skinny_swz_bf16 + aa_swiglu_fwd (hf LlamaMLP :163-176);
      * fused [q | k | v] weight with head_dim 128, mode 'rope128': aa_gemm_skinny_swz_rope_cache_bf16 == GEMV (+ bias) + aa_decode_rope_cache
        (q rotated, k rotated into / v copied into the KV-cache slot of the position) -- with and without the Qwen2 q/k/v bias, MHA and GQA.
    Same operands, same MFMA order, same rounding points -> every bit."""
    from align_anything_amd import ops
    from align_anything_amd.modeling import rope_tables
    for (F, K) in [(64, 128), (11008, 4096), (18944, 3584)]:
        x, wgu = randn_bf16(M, K, seed=1), randn_bf16(2 * F, K, scale=0.05, seed=2)
        want = ops.swiglu_fwd(ops.linear_small(x, ops.SwizzledWeight(wgu)))
        got = ops.gemm_skinny_glu(x, ops.SwizzledWeight(wgu, 'glu'))
        assert got.shape == (M, F) and torch.equal(got, want), (M, F, K, float((got.float() - want.float()).abs().max()))
    for (H, Hkv, K, with_bias, Tmax) in [(4, 4, 256, False, 9), (32, 32, 4096, False, 40), (28, 4, 3584, True, 33)]:
        hd, kw = 128, Hkv * 128
        x, w = randn_bf16(M, K, seed=3), randn_bf16((H + 2 * Hkv) * hd, K, scale=0.05, seed=4)
        bias = randn_bf16((H + 2 * Hkv) * hd, seed=5) if with_bias else None
        cos, sin = rope_tables(64, hd, 1000000.0, dev(), torch.bfloat16)
        pos = torch.randint(0, 64, (M,), generator=torch.Generator().manual_seed(6)).to(torch.int32).to(dev())
        slot = torch.randint(0, Tmax, (M,), generator=torch.Generator().manual_seed(7)).to(dev())
        ca = randn_bf16(M * Tmax, 2 * kw, seed=8)
        cb = ca.clone()
        qkv = ops.linear_small(x, ops.SwizzledWeight(w), bias=bias)
        ops.decode_rope_cache(qkv, H, Hkv, hd, pos, cos, sin, ca, Tmax, slot)
        q = ops.gemm_skinny_rope_cache(x, ops.SwizzledWeight(w, 'rope128'), bias, H, Hkv, pos, cos, sin, cb, Tmax, slot)
        assert torch.equal(q, qkv[:, :H * hd]), (M, H, Hkv, 'q')
        assert torch.equal(ca, cb), (M, H, Hkv, 'cache')


@pytest.mark.parametrize('M', [1, 5, 16])
def test_decode_launch_rules_keep_the_numbers(M):
    """Round 6 (aa_decode_set_rules).  Bit 1 -- software-pipelined trips in the deep 16-wave strips (the down projection, K = 18944 / 11008 / 14336: same k
    order, same accumulator per step parity) -- must give every bit of the plain trips, on row-major and strip-major weights, with the residual epilogue.  Bit 0
    changes the waves per strip of a narrow launch with more strips than CUs (q/k/v of Qwen2-VL-7B: 288 strips; fp32 partial sums are grouped differently):
    equal to the 16-wave launch within fp32 rounding of the accumulator, and both within the usual bound of the fp32 reference."""
    from align_anything_amd import ops
    from tests.util import rel_err
    old = ops.decode_set_rules(0)
    try:
        for (N, K) in [(3584, 18944), (4096, 11008), (4096, 14336), (3584, 8192 + 32)]:
            x, w, res = randn_bf16(M, K, seed=1), randn_bf16(N, K, scale=0.05, seed=2), randn_bf16(M, N, seed=3)
            sw = ops.SwizzledWeight(w)
            got = {}
            for mask in (0, 2):
                ops.decode_set_rules(mask)
                got[mask] = (ops.linear_small(x, w, residual=res), ops.linear_small(x, sw, residual=res))
            assert torch.equal(got[0][0], got[2][0]) and torch.equal(got[0][1], got[2][1]) and torch.equal(got[2][0], got[2][1]), (M, N, K)
            assert rel_err(got[2][0].float(), (x.float() @ w.float().t()).to(torch.bfloat16).float() + res.float()) < 6e-3
        N, K = 4608, 3584          # 288 strips
        x, w, b = randn_bf16(M, K, seed=4), randn_bf16(N, K, scale=0.05, seed=5), randn_bf16(N, seed=6)
        sw = ops.SwizzledWeight(w)
        outs = []
        for mask in (0, 1):
            ops.decode_set_rules(mask)
            outs.append(ops.linear_small(x, sw, bias=b).float())
        ref = x.float() @ w.float().t() + b.float()
        assert rel_err(outs[1], outs[0]) < 2e-3 and rel_err(outs[0], ref) < 6e-3 and rel_err(outs[1], ref) < 6e-3
    finally:
        ops.decode_set_rules(old)


@pytest.mark.parametrize('M', [1, 4, 16])
def test_strip_kernel_with_the_rmsnorm_folded_into_the_weight_copy(M):
    """Round 4: the RMSNorm IN FRONT of a projection folded into the strip kernel (csrc/decode.hip PRO 4): the rollout-only copy holds
    W diag(norm weight) (aa_swizzle_weights_scaled_bf16), the kernel accumulates sum(x^2) from its own fragments and scales the dot products by
    rstd -- for the plain projection (lm_head: + bias / residual exercised too), the [gate; up] copy with the SwiGLU epilogue and the head_dim-128
    q/k/v copy with the rotary + cache-write epilogue.  Yardsticks: fp64 math of norm -> projection (-> epilogue), and the unfolded native chain
    (aa_rmsnorm_fwd + the strip kernel), which the folded kernel must match as closely as that chain matches fp64."""
    from align_anything_amd import ops
    from align_anything_amd.modeling import rope_tables
    eps = 1e-6
    norm64 = lambda x, nw: x.double() * torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + eps) * nw.double()
    for (N, K) in [(64, 128), (1000, 2048), (152064 // 8, 3584), (12288, 4096)]:
        x, w = randn_bf16(M, K, scale=1.5, seed=1), randn_bf16(N, K, scale=0.05, seed=2)
        nw = (1 + 0.2 * torch.randn(K, generator=torch.Generator().manual_seed(3))).to(torch.bfloat16).to(dev())
        bias, res = randn_bf16(N, seed=4), randn_bf16(M, N, seed=5)
        ref = norm64(x, nw) @ w.double().t()
        sw = ops.SwizzledWeight(w, kscale=nw)
        assert sw.folded
        got = ops.linear_small(x, sw, fold_eps=eps)
        unf = ops.linear_small(ops.rmsnorm_fwd(x, nw, eps)[0], ops.SwizzledWeight(w))
        tol = 1e-2 * float(ref.abs().mean()) + 1e-3
        assert_close(got, ref.float(), rtol=1e-2, atol=tol, what=f'folded norm -> strip {M}x{N}x{K}')
        assert float((got.double() - ref).abs().mean()) < 1.5 * float((unf.double() - ref).abs().mean()) + 1e-4
        got = ops.linear_small(x, sw, bias=bias, residual=res, fold_eps=eps)
        assert_close(got, (ref.float() + bias.float()).to(torch.bfloat16).float() + res.float(), rtol=1e-2, atol=tol + 2e-2, what='folded norm -> strip, bias + residual')
        with pytest.raises(RuntimeError):
            ops.linear_small(x, sw)                                    # a folded copy without eps would silently skip the norm
        with pytest.raises(RuntimeError):
            ops.linear_small(x, ops.SwizzledWeight(w), fold_eps=eps)
    for (F, K) in [(64, 128), (18944, 3584)]:
        x, wgu = randn_bf16(M, K, scale=1.5, seed=1), randn_bf16(2 * F, K, scale=0.05, seed=2)
        nw = (1 + 0.2 * torch.randn(K, generator=torch.Generator().manual_seed(3))).to(torch.bfloat16).to(dev())
        gu = norm64(x, nw) @ wgu.double().t()
        ref = torch.nn.functional.silu(gu[:, :F]) * gu[:, F:]
        got = ops.gemm_skinny_glu(x, ops.SwizzledWeight(wgu, 'glu', kscale=nw), eps=eps)
        unf = ops.gemm_skinny_glu(ops.rmsnorm_fwd(x, nw, eps)[0], ops.SwizzledWeight(wgu, 'glu'))
        # silu(gate) * up of two bf16-rounded projections: an element's error scales with the FACTORS (a small product of large factors keeps their
        # absolute error), so the element-wise bound is set by the tensor's rms; the mean-error yardstick below is the sharp one
        tol = 2e-2 * float(ref.pow(2).mean().sqrt()) + 1e-3
        assert_close(got, ref.float(), rtol=2e-2, atol=tol, what=f'folded norm -> glu strip {M}x{F}x{K}')
        assert float((got.double() - ref).abs().mean()) < 1.5 * float((unf.double() - ref).abs().mean()) + 1e-4
    for (H, Hkv, K, with_bias, Tmax) in [(4, 4, 256, False, 9), (28, 4, 3584, True, 33)]:
        hd, kw = 128, Hkv * 128
        x, w = randn_bf16(M, K, scale=1.5, seed=3), randn_bf16((H + 2 * Hkv) * hd, K, scale=0.05, seed=4)
        nw = (1 + 0.2 * torch.randn(K, generator=torch.Generator().manual_seed(3))).to(torch.bfloat16).to(dev())
        bias = randn_bf16((H + 2 * Hkv) * hd, seed=5) if with_bias else None
        cos, sin = rope_tables(64, hd, 1000000.0, dev(), torch.bfloat16)
        pos = torch.randint(0, 64, (M,), generator=torch.Generator().manual_seed(6)).to(torch.int32).to(dev())
        slot = torch.randint(0, Tmax, (M,), generator=torch.Generator().manual_seed(7)).to(dev())
        ca = randn_bf16(M * Tmax, 2 * kw, seed=8)
        cb = ca.clone()
        qa = ops.gemm_skinny_rope_cache(ops.rmsnorm_fwd(x, nw, eps)[0], ops.SwizzledWeight(w, 'rope128'), bias, H, Hkv, pos, cos, sin, ca, Tmax, slot)
        qb = ops.gemm_skinny_rope_cache(x, ops.SwizzledWeight(w, 'rope128', kscale=nw), bias, H, Hkv, pos, cos, sin, cb, Tmax, slot, eps=eps)
        scale = float(qa.float().abs().mean())
        assert_close(qb, qa.float(), rtol=3e-2, atol=3e-2 * scale, what=f'folded norm -> rope strip q {M} {H}/{Hkv}')
        rows = (torch.arange(M, device=dev()) * Tmax + slot)
        assert_close(cb[rows], ca[rows].float(), rtol=3e-2, atol=3e-2 * scale, what='folded norm -> rope strip cache rows')
        other = torch.ones(M * Tmax, dtype=torch.bool, device=dev())
        other[rows] = False
        assert torch.equal(ca[other], cb[other])                         # no other cache slot touched


def test_generate_with_and_without_the_fused_decode_epilogues_gives_the_same_tokens(monkeypatch):
    """The whole rollout with the epilogue fusions on (default) and off (AA_DECODE_EPI=0): identical sequences, greedy and sampled, on a Llama-family
    stack with head_dim 128 (q/k/v bias, GQA) -- the path tools/bench_ppo.py times."""
    from align_anything_amd import configs
    from align_anything_amd.generation import generate
    from align_anything_amd.modeling import build_model
    from bench import random_init_
    text = configs.llama_cfg(512, 1024, 2, 4, 2, 1000, rms_eps=1e-6, rope_theta=1000000.0, head_dim=128, max_position_embeddings=256, attention_bias=True)
    m = build_model(text, 'cuda:0', trainable=False)
    random_init_(m, seed=3, std=0.05)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(3, 1000, (3, 20), generator=g).to(dev())
    mask = torch.ones_like(ids)
    mask[1, :4] = 0
    outs = []
    monkeypatch.setenv('AA_DECODE_NORM_FOLD', '0')           # bit identity is between the UNFOLDED chains; the folded norm is checked below
    for epi in ('1', '0'):
        monkeypatch.setenv('AA_DECODE_EPI', epi)
        greedy = generate(m, ids, mask, max_new_tokens=12, do_sample=False, pad_token_id=0)
        sampled = generate(m, ids, mask, max_new_tokens=12, do_sample=True, temperature=0.8, top_p=0.9, pad_token_id=0,
                           generator=torch.Generator(device=dev()).manual_seed(5))
        modes = {k: v.mode for k, v in m.stack._dw[0].items()}
        assert modes == ({'qkv': 'rope128', 'o': 'plain', 'gu': 'glu', 'down': 'plain'} if epi == '1' else dict.fromkeys(('qkv', 'o', 'gu', 'down'), 'plain'))
        outs.append((greedy.cpu(), sampled.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # default rollout path: RMSNorm folded into the q/k/v, gate/up and lm_head copies (bf16 noise on the logits: greedy tokens may part ways
    # with the unfolded chain at a near-tie, never before the first generated position is compared on real margins in the oracle tests)
    monkeypatch.setenv('AA_DECODE_EPI', '1')
    monkeypatch.setenv('AA_DECODE_NORM_FOLD', '1')
    folded = generate(m, ids, mask, max_new_tokens=12, do_sample=False, pad_token_id=0).cpu()
    assert all(W[k].folded == (k in ('qkv', 'gu')) for W in m.stack._dw for k in W) and m.head._dw.folded
    assert folded.shape == outs[0][0].shape and torch.equal(folded[:, :20], ids.cpu())
    agree = (folded[:, 20:] == outs[0][0][:, 20:]).float().mean()
    assert float(agree) >= 0.5, float(agree)


def test_decode_loop_bookkeeping_kernels_equal_the_torch_ops():
    """aa_decode_record / aa_decode_tick against the torch statements they replace in generation.py (hf GenerationMixin._sample's masking of
    finished rows, the scatter into the output, EosTokenCriteria; the per-row counters): with and without an EOS id, pad == eos included, more
    rows than one block's threads."""
    from align_anything_amd import ops
    g = torch.Generator().manual_seed(0)
    for N, eos, pad in [(1, -1, 0), (4, 7, 0), (16, 7, 7), (700, 3, 1)]:
        Tmax = 12
        sel = torch.randint(0, 10, (N,), generator=g).to(dev())
        unf = (torch.rand(N, generator=g) < 0.7).to(dev())
        out = torch.randint(0, 10, (N, Tmax), generator=g).to(dev())
        tslot = torch.randint(0, Tmax, (N,), generator=g).to(dev())
        nact = torch.tensor([5], device=dev())
        want_out, want_unf = out.clone(), unf.clone()
        want_nact = nact + want_unf.any().to(torch.int64)
        want_tok = torch.where(want_unf, sel, torch.full_like(sel, pad))
        want_out.scatter_(1, tslot[:, None], want_tok[:, None])
        if eos >= 0:
            want_unf.logical_and_(want_tok != eos)
        tok = ops.decode_record(sel, unf, out, tslot, nact, pad, eos)
        assert torch.equal(tok, want_tok) and torch.equal(out, want_out) and torch.equal(unf, want_unf) and torch.equal(nact, want_nact), (N, eos, pad)
        none = torch.zeros(N, dtype=torch.bool, device=dev())
        ops.decode_record(sel, none, out, tslot, nact, pad, eos)                 # every row finished: the step is not counted
        assert torch.equal(nact, want_nact)
        pos, length, step = torch.arange(N, dtype=torch.int32, device=dev()), torch.full((N,), 3, dtype=torch.int32, device=dev()), torch.tensor([9], device=dev())
        t0 = tslot.clone()
        ops.decode_tick(tslot, pos, length, step)
        assert torch.equal(tslot, t0 + 1) and torch.equal(pos, torch.arange(1, N + 1, dtype=torch.int32, device=dev())) and int(length.min()) == 4 == int(length.max()) and int(step) == 10
