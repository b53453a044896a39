"""GPU: SimPO / ORPO / KTO (csrc/pref_losses.hip, trainers/pref.py) against the fixture produced by the reference's
unmodified loss overrides (tests/golden/opt_tiny_pref.npz) and against the oracle restatement."""
import numpy as np
import pytest
import torch

from oracle import rl_math as orl
from tests.gpu_util import assert_close, dev, dump
from tests.util import load_golden, rel_err, state_dict_from_golden, tiny_opt_cfg

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _window_inputs(z):
    """Flat window layout (seq_off) of the padded [2B, W] fixture tensors."""
    resp = [int(r) for r in z['response_lens']]
    off = np.concatenate([[0], np.cumsum([r - 1 for r in resp])]).astype(np.int32)
    lp, rlp = T(z['seq_log_probs']), T(z['ref_seq_log_probs'])
    flat = lambda t: torch.cat([t[i, :resp[i] - 1] for i in range(len(resp))])
    return flat(lp), flat(rlp), torch.from_numpy(off), resp


def test_pair_slice_index_is_bit_exact():
    from align_anything_amd import ops
    z = load_golden('opt_tiny_pref.npz')
    ids, mask = T(z['input_ids']), T(z['attention_mask'])
    _, _, off, resp = _window_inputs(z)
    B = ids.shape[0] // 2
    lo, hi, ln, keep = ops.pair_slice_index(ids.to(dev()), mask.to(dev()), off.to(dev()), B)
    bi, wi = ids.chunk(2); bm, wm = mask.chunk(2)
    for i in range(B):
        sl = orl._pair_slices(bi, wi, bm, wm, i)
        assert bool(keep[i]) == (sl is not None)
        if sl is None:
            continue
        for h, s in ((0, i), (1, i + B)):
            n = resp[s] - 1
            want = range(*sl[h].indices(n))          # python slicing of the row's own window entries
            assert int(lo[s]) - int(off[s]) == (want.start if len(want) else min(sl[h].start, n))
            assert int(hi[s]) - int(lo[s]) == len(want)
            assert int(ln[s]) == sl[2 + h]
    # ragged / degenerate: rows that differ only at the last position, all-equal batch, full-length masks
    g = torch.Generator().manual_seed(1)
    ids2 = torch.randint(3, 99, (1, 300), generator=g).repeat(4, 1)
    ids2[2, 299] = 2
    m2 = torch.ones_like(ids2); m2[1, 200:] = 0
    off2 = torch.tensor([0, 50, 100, 130, 150], dtype=torch.int32)
    lo, hi, ln, keep = ops.pair_slice_index(ids2.to(dev()), m2.to(dev()), off2.to(dev()), 2)
    assert keep.tolist() == [1, 0] and ln.tolist() == [300, 200, 300, 300]
    assert (int(lo[0]), int(hi[0])) == (50, 50) and (int(lo[2]), int(hi[2])) == (130, 130)     # diverge beyond the window: empty


@pytest.mark.parametrize('kind', ['simpo', 'orpo', 'kto'])
def test_pref_loss_kernel_vs_reference_fixture_and_oracle_grad(kind):
    from align_anything_amd import ops
    z = load_golden('opt_tiny_pref.npz')
    ids, mask = T(z['input_ids']), T(z['attention_mask'])
    pol, ref, off, resp = _window_inputs(z)
    B = ids.shape[0] // 2
    beta = float(z['scale_coeff'])
    lo, hi, ln, keep = ops.pair_slice_index(ids.to(dev()), mask.to(dev()), off.to(dev()), B)
    p = {'simpo': (float(z['gamma']), 0, 0), 'orpo': (0, 0, 0), 'kto': (float(z['scale_better']), float(z['scale_worse']), float(z['kto_kl']))}[kind]
    out7, per, dlogp = ops.pref_loss(kind, pol.to(dev()), ref.to(dev()), lo, hi, ln, keep, B, beta, *p)
    kept = keep.bool().cpu()
    assert int(out7[6]) == int(kept.sum()) == 2
    assert abs(float(out7[0]) - float(z[f'{kind}_loss'])) < 2e-6
    assert abs(float(out7[1]) - float(z[f'{kind}_reward_accuracy'])) == 0
    for row, key in ((0, 'better_sample_reward'), (1, 'worse_sample_reward'), (2, 'reward'), (3, 'reward_margin')):
        assert_close(per[row].cpu()[kept], T(z[f'{kind}_{key}']), rtol=1e-5, atol=1e-6, what=f'{kind} {key}')
    # gradient w.r.t. the window log-probs: oracle autograd on the padded tensor
    lp = T(z['seq_log_probs']).clone().requires_grad_(True)
    args = {'simpo': (lp, ids, mask, beta, p[0]), 'orpo': (lp, ids, mask, beta)}.get(kind)
    ld = getattr(orl, kind + '_loss')(*args) if args else orl.kto_loss(lp, T(z['ref_seq_log_probs']), ids, mask, beta, *p)
    ld['loss'].backward()
    want = torch.cat([lp.grad[i, :resp[i] - 1] for i in range(len(resp))])
    assert_close(dlogp.cpu(), want, rtol=1e-5, atol=1e-7, what=f'{kind} dloss/dlogp')
    if kind == 'kto':
        kl = ops.window_kl(pol.to(dev()), ref.to(dev()), pol.numel(), float(len(resp) * (max(resp) - 1)))
        assert abs(float(kl) - float(z['kto_kl'])) < 1e-7


@pytest.mark.parametrize('kind,dtype', [('simpo', 'bf16'), ('orpo', 'bf16'), ('kto', 'bf16'), ('simpo', 'fp32'), ('orpo', 'fp32'), ('kto', 'fp32')])
def test_pref_trainers_match_reference_loss_and_grads(kind, dtype):
    from align_anything_amd.trainers.pref import KTOTrainer, ORPOTrainer, SimPOTrainer
    z, zw = load_golden('opt_tiny_pref.npz'), load_golden('opt_tiny_dpo.npz')
    cls = {'simpo': SimPOTrainer, 'orpo': ORPOTrainer, 'kto': KTOTrainer}[kind]
    cfgs = {'train_cfgs': {'scale_coeff': float(z['scale_coeff']), 'gamma': float(z['gamma']), 'scale_better': float(z['scale_better']),
                           'scale_worse': float(z['scale_worse']), 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant',
                           'weight_decay': 0.0, 'compute_dtype': dtype},
            'model_cfgs': {'pad_token_id': int(z['pad_token_id'])}}
    tr = cls(cfgs, {'gradient_clipping': 1.0}, model_cfg=tiny_opt_cfg(), policy_state=state_dict_from_golden(zw, 'w.'),
             reference_state=state_dict_from_golden(zw, 'r.'), device='cuda:0')
    assert (tr.reference is None) == (kind != 'kto')
    b = {'input_ids': T(z['input_ids']).to(dev()), 'attention_mask': T(z['attention_mask']).to(dev()),
         'meta_info': {'response_lens': [int(r) for r in z['response_lens']]}}
    tight = dtype == 'fp32'
    if kind == 'kto':
        kl = tr.compute_kl(b)
        assert abs(float(kl) - float(z['kto_kl'])) < (1e-6 if tight else 2e-3)
        tr.kl = float(z['kto_kl'])          # identical shift on both sides for the loss comparison
    ld = tr.loss(b)
    rep = [f"{kind}/{dtype}: loss native {float(ld['loss']):.6f} reference {float(z[kind + '_loss']):.6f}"]
    assert abs(float(ld['loss']) - float(z[f'{kind}_loss'])) < (2e-5 if tight else 2e-2)
    assert ld['reward'].shape == tuple(z[f'{kind}_reward'].shape)
    assert_close(ld['reward_margin'].cpu(), T(z[f'{kind}_reward_margin']), rtol=1e-2, atol=(2e-5 if tight else 3e-2), what='margin')
    tr.model.backward(ld['loss'])
    torch.cuda.synchronize()
    for k in z.files:
        if k.startswith(kind + '_g.'):
            n = k[len(kind) + 3:]
            e = rel_err(tr.policy.store.grad_view(n).float().cpu().reshape(z[k].shape), T(z[k]))
            rep.append(f'  grad {n}: rel_err {e:.2e}')
            assert e < (2e-4 if tight else 8e-2), (n, e)
    info = tr.train_step(b)
    assert np.isfinite(info['train/loss'])
    dump(f'parity_{kind}_{dtype}.txt', '\n'.join(rep) + '\n')
