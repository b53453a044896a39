"""GPU: GRPO kernels (aa_group_advantage / aa_completion_mask / aa_grpo_loss_fwd_bwd) and the native
GRPOTrainer.train_step against the oracle and against what the reference's unmodified
align_anything/trainers/text_to_text/grpo.py::GRPOTrainer.train_step produced (tests/golden/grpo_tiny.npz)."""
import numpy as np
import pytest
import torch

from oracle import rl_math as orl
from tests.gpu_util import assert_close, dev, dump
from tests.util import load_golden, rel_err, state_dict_from_golden, tiny_opt_cfg

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.mark.parametrize('B,G,L', [(2, 3, 14), (5, 8, 257), (1, 2, 1)])
def test_grpo_kernels_vs_oracle(B, G, L):
    from align_anything_amd import ops
    g = torch.Generator().manual_seed(B * 100 + L)
    rows, EOS = B * G, 2
    logp = (-torch.rand(rows, L, generator=g) * 4).requires_grad_(True)
    ref = -torch.rand(rows, L, generator=g) * 4
    rewards = torch.randn(rows, generator=g) * 3
    toks = torch.randint(0, 12, (rows, L), generator=g)          # small vocab -> many rows contain EOS, some do not
    toks[0] = 5                                                  # a row with no EOS at all
    if L > 1:
        toks[rows - 1, 0] = EOS                                  # EOS at the very first completion token
    o_loss, o_adv, o_mask = orl.grpo_loss(logp, ref, rewards, B, G, toks, EOS, 0.04)
    o_loss.backward()
    d = lambda t: t.detach().to(dev())
    adv = ops.group_advantage(d(rewards), B, G)
    mask = ops.completion_mask(d(toks), EOS)
    assert torch.equal(mask.cpu().bool(), o_mask.bool()), 'completion mask is integer work: bit-exact'
    assert_close(adv.cpu(), o_adv, rtol=1e-5, atol=1e-5, what='group advantage')
    loss, dlogp = ops.grpo_loss(d(logp), d(ref), adv, mask, 0.04)
    assert_close(loss.cpu().reshape(()), o_loss.detach(), rtol=1e-5, atol=1e-6, what='grpo loss')
    assert_close(dlogp.cpu(), logp.grad, rtol=1e-5, atol=1e-7, what='grpo dloss/dlogp')
    # strided completion view (sequences[:, P:]) must give the same mask
    wide = torch.cat([torch.full((rows, 3), EOS), toks], 1).to(dev())
    assert torch.equal(ops.completion_mask(wide[:, 3:], EOS), mask)


def _trainer(z, **train_cfgs):
    from align_anything_amd.trainers.grpo import GRPOTrainer
    cfg = tiny_opt_cfg()
    cfgs = {'train_cfgs': dict({'actor_lr': 1e-3, 'actor_weight_decay': 0.0, 'actor_lr_warmup_ratio': 0.0,
                                'actor_lr_scheduler_type': 'constant', 'beta': float(z['beta']), 'num_generations': int(z['G'])},
                               **train_cfgs),
            'model_cfgs': {'pad_token_id': int(z['pad']), 'eos_token_id': int(z['eos']), 'model_max_length': 24,
                           'temperature': 1.0, 'top_p': 1.0}}
    actor_sd = state_dict_from_golden(z, 'w.', torch.bfloat16)
    ref_sd = state_dict_from_golden(z, 'r.', torch.bfloat16)
    g = torch.Generator().manual_seed(4)
    rm_sd = {k: v for k, v in actor_sd.items() if k != 'lm_head.weight'}
    rm_sd['score_head.weight'] = (torch.randn(1, cfg['hidden_size'], generator=g) * 0.2).to(torch.bfloat16)
    return GRPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, actor_state=actor_sd, reference_state=ref_sd,
                       reward_state=rm_sd, device='cuda:0')


def test_grpo_train_step_matches_reference_golden():
    z = load_golden('grpo_tiny.npz')
    tr = _trainer(z)
    prompts = T(z['prompts'])
    pb = {'input_ids': prompts, 'attention_mask': torch.ones_like(prompts)}
    seqs, rewards = T(z['sequences']).to(dev()), T(z['rewards']).to(dev())
    P, K = prompts.shape[1], z['sequences'].shape[1] - prompts.shape[1]
    am = (seqs != int(z['pad'])).long()
    lp, _ = tr._get_per_token_logps(tr.actor_model, seqs, am, K)
    rlp, _ = tr._get_per_token_logps(tr.actor_reference_model, seqs, am, K)
    keep = T(orl.grpo_loss(T(z['per_token_logps']), T(z['ref_per_token_logps']), T(z['rewards']), int(z['B']), int(z['G']),
                           T(z['sequences'])[:, P:], int(z['eos']), 0.04)[2].numpy()).bool()
    # bf16 forward vs the reference's fp32 forward, on the tokens the loss counts
    assert_close(lp.cpu()[keep], T(z['per_token_logps'])[keep], rtol=2e-2, atol=6e-2, what='actor per-token logps')
    assert_close(rlp.cpu()[keep], T(z['ref_per_token_logps'])[keep], rtol=2e-2, atol=6e-2, what='ref per-token logps')
    info = tr.train_step(pb, sequences=seqs, rewards=rewards)
    rep = [f"loss native {info['train/loss']:.6f} reference {float(z['loss']):.6f}",
           f"reward native {info['train/reward']:.6f} reference {float(z['reward_mean']):.6f}"]
    assert abs(info['train/loss'] - float(z['loss'])) < 1e-2
    assert abs(info['train/reward'] - float(z['reward_mean'])) < 1e-5
    tr.actor_model.wait_optimizer()
    torch.cuda.synchronize()
    for k in z.files:
        if k.startswith('g.'):
            n = k[2:]
            got = tr.actor_model.module.store.grad_view(n).float().cpu().reshape(z[k].shape)
            e = rel_err(got, T(z[k]))
            rep.append(f'{n}: rel_err {e:.4f}')
            assert e < 8e-2, (n, e)
    dump('parity_grpo_train_step.txt', '\n'.join(rep) + '\n')


def test_grpo_full_step_generates_scores_and_updates():
    """End to end: sample G completions per prompt, score them with the native reward model, update."""
    z = load_golden('grpo_tiny.npz')
    tr = _trainer(z)
    prompts = T(z['prompts']).to(dev())
    pb = {'input_ids': prompts, 'attention_mask': torch.ones_like(prompts)}
    gen = torch.Generator(device='cuda').manual_seed(0)
    seqs = tr.generate_completions(pb, gen)
    B, G, P = prompts.shape[0], int(z['G']), prompts.shape[1]
    assert seqs.shape[0] == B * G and seqs.shape[1] <= 24
    assert torch.equal(seqs[:, :P], prompts.repeat_interleave(G, 0))       # row b*G+g belongs to prompt b
    assert len({tuple(r.tolist()) for r in seqs[:G, P:]}) > 1              # sampling: group members differ
    rewards = tr.compute_rewards(seqs, P)
    assert rewards.shape == (B * G,) and torch.isfinite(rewards).all()
    # reward of a row must not depend on what follows its first EOS
    s2 = seqs.clone()
    s2[0, P + 2] = int(z['eos'])
    r_a = tr.compute_rewards(s2, P)
    s2[0, P + 3:] = 7
    r_b = tr.compute_rewards(s2, P)
    assert torch.equal(r_a, r_b)
    w0 = tr.actor_model.module.store.view('model.decoder.layers.0.fc1.weight').float().clone()
    info = tr.train_step(pb, generator=gen)
    tr.actor_model.wait_optimizer()
    torch.cuda.synchronize()
    assert np.isfinite(info['train/loss']) and np.isfinite(info['train/reward'])
    assert (tr.actor_model.module.store.view('model.decoder.layers.0.fc1.weight').float() - w0).abs().max() > 0
