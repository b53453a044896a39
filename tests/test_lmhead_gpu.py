"""GPU: lm_head x log-prob without the logits buffer (csrc/lmhead.hip) vs the unfused pair of kernels, the oracle and the
reference's golden vectors (utils/tools.py:402-413 applied to `model(**batch).logits`, dpo.py:128-138)."""
import pytest
import torch

from oracle import rl_math as orl
from tests.gpu_util import assert_close, dev, randn_bf16
from tests.util import bits_to_bf16, load_golden

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _case(rows, V, h, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    hidden = (torch.randn(rows, h, generator=g) * 1.0).to(dtype).to(dev())
    w = (torch.randn(V, h, generator=g) * (2.0 / h ** 0.5)).to(dtype).to(dev())
    labels = torch.randint(0, V, (rows,), generator=g)
    labels[0], labels[1] = V - 1, 0          # first / last column, last chunk
    return hidden, w, labels.to(dev())


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('rows,V,h,chunk', [(128, 32064, 512, 8192), (64, 32064, 256, 2048), (192, 6144, 256, 2048), (64, 4096, 128, 8192)])
def test_forward_is_bit_identical_to_gemm_plus_logprob_gather(dtype, rows, V, h, chunk):
    from align_anything_amd import ops
    hidden, w, labels = _case(rows, V, h, dtype, seed=V + h)
    logits = ops.gemm(hidden, w)
    lp0, lse0 = ops.logprob_gather_fwd(logits, labels, round_bf16=(dtype == torch.bfloat16))
    lp1, lse1 = ops.lmhead_logprob_fwd(hidden, w, labels, round_bf16=(dtype == torch.bfloat16), chunk=chunk)
    assert torch.equal(lse1, lse0) and torch.equal(lp1, lp0)
    # and against the oracle's log_softmax/gather on the same (rounded) logits
    ref = orl.gather_log_probabilities(logits.float().cpu()[None], labels.cpu()[None])[0]
    lp32, _ = ops.lmhead_logprob_fwd(hidden, w, labels, chunk=chunk)
    assert_close(lp32.cpu(), ref, rtol=1e-5, atol=3e-5, what='logp vs oracle')


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('rows,V,h,chunk', [(128, 32064, 512, 8192), (64, 6144, 256, 2048)])
def test_backward_matches_the_unfused_kernels(dtype, rows, V, h, chunk):
    from align_anything_amd import ops
    hidden, w, labels = _case(rows, V, h, dtype, seed=V)
    dlogp = torch.randn(rows, generator=torch.Generator().manual_seed(5)).to(dev())
    logits = ops.gemm(hidden, w)
    _, lse = ops.logprob_gather_fwd(logits, labels)
    dlog = ops.logprob_gather_bwd(logits, labels, lse, dlogp)
    dh0 = ops.gemm(dlog, w, b_n=True)
    gdt = torch.float32 if dtype == torch.float32 else torch.bfloat16
    dw0 = torch.zeros(V, h, dtype=gdt, device=dev())
    ops.gemm(dlog, hidden, out=dw0, a_t=True, b_n=True)
    dw1 = torch.full((V, h), 7.0, dtype=gdt, device=dev())       # overwritten, not accumulated
    dh1 = ops.lmhead_logprob_bwd(hidden, w, labels, lse, dlogp, dw=dw1, chunk=chunk)
    # dW: every [chunk, h] block is the same K-loop over the rows -> bit-identical
    assert torch.equal(dw1, dw0)
    # d_hidden: fp32 accumulation across chunks instead of one K-loop over V -> the same value up to the rounding of the result
    tol = 8e-3 if dtype == torch.bfloat16 else 2e-5
    assert_close(dh1, dh0, rtol=tol, atol=tol * float(dh0.float().abs().max()) * 0.1, what='d_hidden')
    if dtype == torch.bfloat16:
        frac_equal = (dh1 == dh0).float().mean().item()
        assert frac_equal > 0.98, frac_equal
    # accumulate into an fp32 gradient buffer
    dw2 = torch.ones(V, h, dtype=torch.float32, device=dev())
    ops.lmhead_logprob_bwd(hidden, w, labels, lse, dlogp, dw=dw2, accumulate=True, chunk=chunk)
    dw_ref = torch.ones(V, h, dtype=torch.float32, device=dev())
    ops.gemm(dlog, hidden, out=dw_ref, a_t=True, b_n=True, accumulate=True)
    assert torch.equal(dw2, dw_ref)
    # no weight gradient requested (reference model / frozen head)
    dh3 = ops.lmhead_logprob_bwd(hidden, w, labels, lse, dlogp, dw=None, chunk=chunk)
    assert torch.equal(dh3, dh1)


def test_golden_vectors_through_an_identity_lm_head():
    """The reference's gather_log_probabilities fixture (tests/golden/rl_math.npz): the logits are fed as hidden states
    through an identity lm_head, so the fused path sees exactly the fixture's logits."""
    from align_anything_amd import ops
    z = load_golden('rl_math.npz')
    labels = torch.zeros(64, dtype=torch.int64)
    labels[:37] = T(z['glp_labels'])
    labels = labels.to(dev())
    V, h = 1000, 1024
    eye = torch.zeros(V, h)
    eye[torch.arange(V), torch.arange(V)] = 1.0
    for dtype, key, out, tol in ((torch.float32, 'glp_logits', 'glp_out_f32', dict(rtol=1e-5, atol=2e-5)),
                                 (torch.bfloat16, 'glp_logits_bf16', 'glp_out_bf16', dict(rtol=8e-3, atol=0))):
        lg = T(z[key]) if dtype == torch.float32 else bits_to_bf16(z[key])
        hidden = torch.zeros(64, h, dtype=dtype)
        hidden[:37, :V] = lg
        lp, _ = ops.lmhead_logprob_fwd(hidden.to(dev()), eye.to(dtype).to(dev()), labels, round_bf16=(dtype == torch.bfloat16), chunk=2048)
        assert_close(lp[:37], T(z[out]).to(dev()), what=f'golden {out}', **tol)
        lp_unfused, _ = ops.logprob_gather_fwd(lg.to(dev()), labels[:37], round_bf16=(dtype == torch.bfloat16))
        assert torch.equal(lp[:37], lp_unfused)


def test_bad_arguments_fail_loudly():
    from align_anything_amd import ops
    from align_anything_amd.lib import AAHipError
    hidden, w, labels = _case(64, 4096, 128, torch.bfloat16)
    with pytest.raises(AAHipError, match='multiple of 2048'):
        ops.lmhead_logprob_fwd(hidden, w, labels, chunk=1000)
    with pytest.raises(RuntimeError):
        ops.lmhead_logprob_fwd(hidden, w.float(), labels)
    labels_bad = labels.clone(); labels_bad[3] = 4096        # torch.gather would raise; the kernels surface NaN
    lp, _ = ops.lmhead_logprob_fwd(hidden, w, labels_bad)
    assert torch.isnan(lp[3]) and not torch.isnan(lp[:3]).any()


def test_policy_step_same_with_and_without_the_logits_buffer(monkeypatch):
    """DPOTrainer.loss + backward on the reference's LLaVA fixture (V=320 = one ragged chunk) and on OPT's tied lm_head: AA_LMHEAD_FUSED=0
    (materialised logits) and the default give the same loss bit for bit and the same gradients up to d_hidden's one rounding."""
    from tests.test_model_gpu import _batch, _trainer
    from tests.util import tiny_llava_cfg, tiny_opt_cfg
    for fixture, cfg, pix in (('llava_tiny_dpo.npz', tiny_llava_cfg(), True), ('opt_tiny_dpo.npz', tiny_opt_cfg(), False)):
        z = load_golden(fixture)
        out = {}
        for fused in ('1', '0'):
            monkeypatch.setenv('AA_LMHEAD_FUSED', fused)
            tr = _trainer(z, cfg)
            ld = tr.loss(_batch(z, with_pixels=pix))
            tr.model.backward(ld['loss'])
            torch.cuda.synchronize()
            st = tr.policy.store
            out[fused] = (ld['loss'].clone(), {n: g.float().clone() for n, g in st.g.items()})
        assert torch.equal(out['1'][0], out['0'][0]), fixture
        for k, g1 in out['1'][1].items():
            g0 = out['0'][1][k]
            assert_close(g1, g0, rtol=2e-2, atol=2e-2 * float(g0.abs().max()) + 1e-8, what=f'{fixture} {k}')


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('rows,V,h,chunk', [(64, 50272, 768, 8192), (64, 2048 + 40, 128, 2048)])
def test_ragged_vocabulary_backward(dtype, rows, V, h, chunk):
    """OPT's vocabulary (50272 = 64 x 785 + 32) is not a multiple of the GEMM's contraction granule: the last piece of
    d_hidden = dlogits @ W runs against a zero-padded copy of the tail rows.  Checked against plain fp32 torch math on the
    same operands (tolerance: bf16 rounding of the logits / dlogits, 1e-5 relative for the fp32 twin)."""
    from align_anything_amd import ops
    hidden, w, labels = _case(rows, V, h, dtype, seed=V)
    dlogp = torch.randn(rows, generator=torch.Generator().manual_seed(7)).to(dev())
    lp, lse = ops.lmhead_logprob_fwd(hidden, w, labels, chunk=chunk)
    gdt = torch.float32
    dw = torch.zeros(V, h, dtype=gdt, device=dev())
    dh = ops.lmhead_logprob_bwd(hidden, w, labels, lse, dlogp, dw=dw, chunk=chunk)
    H, W = hidden.float(), w.float()
    logits = H @ W.t()
    if dtype == torch.bfloat16:
        logits = logits.bfloat16().float()           # the kernels round the logits chunk to bf16 like the reference's bf16 lm_head
    ref_lp = torch.log_softmax(logits, -1).gather(1, labels[:, None])[:, 0]
    dlog = -torch.softmax(logits, -1) * dlogp[:, None]
    dlog[torch.arange(rows), labels] += dlogp
    if dtype == torch.bfloat16:
        dlog = dlog.bfloat16().float()
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    assert_close(lp, ref_lp, rtol=tol, atol=tol, what='logp')
    ref_dh, ref_dw = dlog @ W, dlog.t() @ H
    assert_close(dh.float(), ref_dh, rtol=tol, atol=tol * float(ref_dh.abs().max()), what='d_hidden (ragged V)')
    assert_close(dw, ref_dw, rtol=tol, atol=tol * float(ref_dw.abs().max()), what='dW (ragged V)')
