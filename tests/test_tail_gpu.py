"""Dead-row elimination in the last decoder layer (modeling.LlamaStack.forward `tail`): the log-prob head reads the response-window rows only
(dpo.py:128-136 slices the logits), so after the last layer's keys and values nothing consumes the other rows -- that layer's queries, attention output,
o-projection and MLP run on the window rows alone.  Every consumed number is computed by the same kernels on the same operands: log-probs and loss must be
BIT-identical to the full computation (AA_TAIL_PRUNE=0), gradients equal up to the order of fp32 partial sums in that layer's weight gradients."""
import pytest
import torch

from tests.gpu_util import dev, dump
from tests.test_pack_gpu import _pair_batch
from tests.util import load_golden, rel_err, state_dict_from_golden, tiny_llava_cfg

pytestmark = pytest.mark.gpu


def _poison_unwritten_attention_rows(monkeypatch):
    """The attention output of a skipped query block is never written.  Whatever reads it by mistake (a pad row of the window gather did, before its pad entries
    pointed at a window row) must meet NaN, not the finite leftovers of the allocator that hide it: every attention output buffer the models allocate starts as NaN."""
    from align_anything_amd import ops
    real = ops.attn_fwd

    def attn_fwd(q, k, v, N, T, H, Hkv, hd, *a, **kw):
        if kw.get('out') is None:
            kw['out'] = torch.full((q.shape[0], H * hd), float('nan'), dtype=q.dtype, device=q.device)
        return real(q, k, v, N, T, H, Hkv, hd, *a, **kw)
    monkeypatch.setattr(ops, 'attn_fwd', attn_fwd)


def _step(dtype, prune, monkeypatch, share=False):
    from align_anything_amd import modeling
    from align_anything_amd.trainers.dpo import DPOTrainer
    monkeypatch.setattr(modeling, 'TAIL_PRUNE', prune)
    z = load_golden('llava_tiny_dpo.npz')
    cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-4, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant', 'compute_dtype': dtype, 'share_prompt_prefix': share},
            'model_cfgs': {'pad_token_id': 301}}
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=tiny_llava_cfg(), policy_state=state_dict_from_golden(z, 'w.', torch.bfloat16),
                    reference_state=state_dict_from_golden(z, 'r.', torch.bfloat16), device='cuda:0')
    b = _pair_batch(2, 256, (150, 90), (64, 100), (30, 77), seed=11)
    _poison_unwritten_attention_rows(monkeypatch)
    lp = tr.compute_log_probs(tr.model, b)
    used = tr.policy.stack.tail_used
    rlp = tr.compute_log_probs(tr.reference_model, b)
    ld = tr.loss(b)
    tr.model.backward(ld['loss'])
    torch.cuda.synchronize()
    st = tr.policy.store
    return lp.float().cpu(), rlp.float().cpu(), float(ld['loss']), {n: st.grad_view(n).float().clone() for n in st.hf_names() if st.grad_view(n) is not None}, used


@pytest.mark.parametrize('share', [False, True])
@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_last_layer_on_the_window_rows_only_changes_no_consumed_number(dtype, share, monkeypatch):
    """share: on top of shared-prompt packing (the stack's rows are then the packed token rows; the attention output is still read by [N, T] slot)."""
    lp0, rlp0, l0, g0, u0 = _step(dtype, False, monkeypatch, share)
    lp1, rlp1, l1, g1, u1 = _step(dtype, True, monkeypatch, share)
    assert u1 and not u0
    assert torch.equal(lp0, lp1) and torch.equal(rlp0, rlp1) and l0 == l1
    assert all(bool(torch.isfinite(g1[n]).all()) for n in g1)
    worst = max((rel_err(g1[n], g0[n]), n) for n in g0 if float(g0[n].norm()) > 1e-6)
    last = [n for n in g0 if '.layers.1.' in n and 'language_model' in n]
    assert last and len(g0) > 20
    dump(f'parity_tail_prune_{dtype}{"_packed" if share else ""}.txt', f'{dtype}{" + shared-prompt packing" if share else ""}: log-probs / loss bit-identical with and without the last-layer dead-row elimination; worst gradient rel_err {worst[0]:.2e} ({worst[1]}) '
         f'over {len(g0)} tensors\n')
    assert worst[0] < (2e-6 if dtype == 'fp32' else 4e-3), worst


def test_other_consumers_of_the_stack_keep_every_row(monkeypatch):
    """`logits()` (all positions), the score heads and the KV-cache prefill read rows outside the windows: the switch must not reach them."""
    from align_anything_amd import modeling
    from align_anything_amd.trainers.dpo import DPOTrainer
    monkeypatch.setattr(modeling, 'TAIL_PRUNE', True)
    z = load_golden('llava_tiny_dpo.npz')
    cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-4, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant'}, 'model_cfgs': {'pad_token_id': 301}}
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=tiny_llava_cfg(), policy_state=state_dict_from_golden(z, 'w.', torch.bfloat16),
                    reference_state=state_dict_from_golden(z, 'r.', torch.bfloat16), device='cuda:0')
    b = _pair_batch(2, 256, (150, 90), (64, 100), (30, 77), seed=11)
    tr.compute_log_probs(tr.model, b)
    assert tr.policy.stack.tail_used and tr.policy.stack.tail is None
    full = tr.policy.logits(b['input_ids'], b['attention_mask'], pixel_values=b['pixel_values'])
    assert not tr.policy.stack.tail_used and full.shape[:2] == b['input_ids'].shape and bool(torch.isfinite(full.float()).all())


@pytest.mark.parametrize('share', [False, True])
def test_last_moe_layer_on_the_window_rows_only(share, monkeypatch):
    """Qwen3MoeStack: the last layer's attention queries, o-projection, ROUTER and experts run on the response-window rows only -- a token's routing and its
    expert outputs do not depend on which other rows share its tile, so log-probs stay bit-identical (fp32 twin; also on top of shared-prompt packing)."""
    from align_anything_amd import modeling
    from tests.test_qwen3moe_gpu import _trainer
    z = load_golden('qwen3moe_tiny_dpo.npz')
    out = {}
    for prune in (False, True):
        monkeypatch.setattr(modeling, 'TAIL_PRUNE', prune)
        tr = _trainer(z, 'fp32')
        tr.share_prompt_prefix = share
        tr.pad_token_id = 301
        b = _pair_batch(2, 224, (120, 90), (40, 70), (90, 25), 6, 0)
        b.pop('pixel_values')
        _poison_unwritten_attention_rows(monkeypatch)
        lp = tr.compute_log_probs(tr.model, b).float().cpu()
        used = tr.policy.stack.tail_used
        ld = tr.loss(b)
        tr.model.backward(ld['loss'])
        torch.cuda.synchronize()
        st = tr.policy.store
        out[prune] = (lp, float(ld['loss']), {n: st.grad_view(n).float().clone() for n in st.hf_names() if st.grad_view(n) is not None}, used, b.get('_pack'))
    (lp0, l0, g0, u0, p0), (lp1, l1, g1, u1, p1) = out[False], out[True]
    assert u1 and not u0 and (p1 is not None) == share
    assert torch.equal(lp0, lp1) and l0 == l1
    assert all(bool(torch.isfinite(g1[n]).all()) for n in g1)
    worst = max((rel_err(g1[n], g0[n]), n) for n in g0 if float(g0[n].norm()) > 1e-6)
    dump(f'parity_tail_prune_qwen3moe{"_packed" if share else ""}.txt', f'fp32{" + shared-prompt packing" if share else ""}: log-probs / loss bit-identical; worst gradient rel_err {worst[0]:.2e} ({worst[1]}) over {len(g0)} tensors\n')
    assert worst[0] < 5e-6, worst


@pytest.mark.parametrize('share', [False, True])
def test_no_kernel_of_the_step_reads_uninitialised_memory(share, monkeypatch):
    """torch's deterministic mode fills every `torch.empty` with NaN (torch.utils.deterministic.fill_uninitialized_memory).  A whole DPO step under it -- default
    path and shared-prompt packing, dead-row elimination on -- must give the bits of the ordinary run: no kernel reads a row it (or a predecessor) did not write."""
    outs = []
    for fill in (False, True):
        if fill:
            torch.use_deterministic_algorithms(True, warn_only=True)
            torch.utils.deterministic.fill_uninitialized_memory = True
            assert bool(torch.isnan(torch.empty(4096, device=dev())).all()) and bool(torch.isnan(torch.empty((64, 64), dtype=torch.bfloat16, device=dev()).float()).all())
        try:
            lp, rlp, loss, grads, used = _step('bf16', True, monkeypatch, share)
        finally:
            torch.use_deterministic_algorithms(False)
        outs.append((lp, rlp, loss, grads))
        assert used
    (lp0, rlp0, l0, g0), (lp1, rlp1, l1, g1) = outs
    assert l0 == l1 and torch.equal(lp0, lp1) and torch.equal(rlp0, rlp1)
    assert all(bool(torch.isfinite(g1[n]).all()) and torch.equal(g0[n], g1[n]) for n in g0)
