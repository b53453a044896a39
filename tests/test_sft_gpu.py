"""GPU: native supervised fine-tuning step (trainers/sft.py::SupervisedTrainer, `aa_sft_loss_fwd_bwd`) against the fixture the
reference's own SupervisedTrainer.loss produced on HF OPTForCausalLM (tests/golden/opt_tiny_sft.npz: right-padded rows, prompt and
padding labels -100): loss and every gradient."""
import numpy as np
import pytest
import torch

from tests.gpu_util import dev, dump
from tests.util import load_golden, rel_err, state_dict_from_golden, tiny_opt_cfg

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_sft_step_matches_reference_fixture(dtype):
    from align_anything_amd import ops
    from align_anything_amd.trainers.sft import SupervisedTrainer
    z, zw = load_golden('opt_tiny_sft.npz'), load_golden('opt_tiny_dpo.npz')
    tight = dtype == 'fp32'
    cfgs = {'train_cfgs': {'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant', 'weight_decay': 0.0, 'compute_dtype': dtype},
            'model_cfgs': {'pad_token_id': 1}}
    wd = torch.float32 if tight else torch.bfloat16
    tr = SupervisedTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=tiny_opt_cfg(), policy_state=state_dict_from_golden(zw, 'w.', wd), device='cuda:0')
    assert tr.reference is None and tr.reference_model is None
    b = {'input_ids': T(z['input_ids']).to(dev()), 'attention_mask': T(z['attention_mask']).to(dev()), 'labels': T(z['labels'])}
    ld = tr.loss(b)
    assert abs(float(ld['loss']) - float(z['loss'])) < (3e-5 if tight else 2e-2), (float(ld['loss']), float(z['loss']))
    tr.model.backward(ld['loss'])
    torch.cuda.synchronize()
    worst, n = 0.0, 0
    gots, wants = [], []
    big = max(float(T(z[k]).norm()) for k in z.files if k.startswith('g.'))
    for k in z.files:
        if not k.startswith('g.'):
            continue
        g = tr.policy.store.grad_view(k[2:])
        assert g is not None, k
        want = T(z[k])
        got = g.float().cpu().reshape(want.shape)
        gots.append(got.reshape(-1)); wants.append(want.reshape(-1))
        if float(want.norm()) < 1e-6:          # structurally zero (k_proj bias: a shift of every key leaves the softmax unchanged)
            assert float(got.norm()) < (1e-4 if tight else 5e-3), (k, float(got.norm()))     # bf16: rounding noise of O(1) cross-entropy gradients
            continue
        e = rel_err(got, want)
        n += 1
        if tight or float(want.norm()) > 1e-2 * big:       # bf16: per-tensor bound on the tensors that carry the gradient, global bound on all
            worst = max(worst, e)
            assert e < (5e-4 if tight else 2.5e-1), (k, e)
    e_all = rel_err(torch.cat(gots), torch.cat(wants))
    assert e_all < (1e-5 if tight else 1.5e-1), e_all          # a pure-bf16 CPU emulation of the same step sits at 5e-2 (worst major tensor 9e-2)
    assert n >= 20
    dump(f'parity_sft_{dtype}.txt', f'{dtype}: loss native {float(ld["loss"]):.6f} reference {float(z["loss"]):.6f}; worst gradient rel_err {worst:.2e} over {n} tensors, all gradients together {e_all:.2e}\n')
    info = tr.train_step(b)
    assert np.isfinite(info['train/loss']) and abs(info['train/loss'] - float(ld['loss'])) < 1e-6 and info['train/lr'] == 1e-3
    # kernel alone: -mean over the real rows, zero gradient on the padding rows
    lp = torch.randn(128, device=dev())
    loss, dl = ops.sft_loss(lp, 70)
    assert abs(float(loss) + float(lp[:70].mean())) < 1e-6 and torch.allclose(dl[:70], torch.full((70,), -1 / 70, device=dev())) and float(dl[70:].abs().sum()) == 0.0
