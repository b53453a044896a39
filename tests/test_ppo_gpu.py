"""GPU: native PPO update phase (trainers/ppo.py::PPOTrainer.rl_step, reward_model_step) against the CPU oracle:
reference math (oracle/rl_math.py, pinned to the reference's golden vectors) on the oracle OPT model with
autograd.  Mirrors align_anything/trainers/text_to_text/ppo.py:224-242, 309-398."""
import pytest
import torch

from oracle import models as om
from oracle import rl_math as orl
from tests.gpu_util import assert_close, dev, dump
from tests.util import load_golden, rel_err, state_dict_from_golden, tiny_opt_cfg

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _setup():
    from align_anything_amd.trainers.ppo import PPOTrainer
    z = load_golden('opt_tiny_dpo.npz')
    cfg = tiny_opt_cfg()
    g = torch.Generator().manual_seed(21)
    actor_sd = state_dict_from_golden(z, 'w.', torch.bfloat16)
    old_sd = state_dict_from_golden(z, 'r.', torch.float32)          # "old policy" that produced the rollout
    score_w = (torch.randn(1, cfg['hidden_size'], generator=g) * 0.2).to(torch.bfloat16)
    rm_sd = {k: v for k, v in actor_sd.items() if k != 'lm_head.weight'}
    rm_sd['score_head.weight'] = score_w
    cfgs = {'train_cfgs': {'actor_lr': 1e-3, 'critic_lr': 1e-3, 'actor_weight_decay': 0.0, 'critic_weight_decay': 0.0,
                           'actor_lr_warmup_ratio': 0.0, 'critic_lr_warmup_ratio': 0.0, 'actor_lr_scheduler_type': 'constant',
                           'critic_lr_scheduler_type': 'constant', 'kl_coeff': 0.02, 'clip_range_ratio': 0.2,
                           'clip_range_value': 5.0, 'clip_range_score': 50.0, 'gamma': 1.0, 'gae_lambda': 0.95}}
    tr = PPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, actor_state=actor_sd, reward_state=rm_sd, device='cuda:0')
    # rollout-shaped batch: left-padded prompt (start = 12 prompt positions) + response, right padding after "EOS"
    N, Tn, start = 3, 40, 12
    ids = torch.randint(3, 320, (N, Tn), generator=g)
    mask = torch.ones(N, Tn, dtype=torch.long)
    for n, (lp, rp) in enumerate(((0, 0), (5, 7), (2, 15))):
        mask[n, :lp] = 0; ids[n, :lp] = 1
        if rp:
            mask[n, Tn - rp:] = 0; ids[n, Tn - rp:] = 1
    return tr, z, cfg, actor_sd, old_sd, rm_sd, ids, mask, start


def test_reward_model_step_and_rl_step_match_oracle():
    tr, z, cfg, actor_sd, old_sd, rm_sd, ids, mask, start = _setup()
    # ---------------- oracle side (fp32 CPU)
    f32 = lambda sd: {k: v.float().clone().requires_grad_(True) for k, v in sd.items()}
    a_sd = f32({k: v for k, v in actor_sd.items() if k != 'lm_head.weight'})
    c_sd = f32(rm_sd)
    with torch.no_grad():
        old_lp = orl.gather_log_probabilities(om.opt_logits(old_sd, cfg, ids, mask)[:, :-1], ids[:, 1:])
        ref_lp = orl.gather_log_probabilities(om.opt_logits({k: v.detach() for k, v in a_sd.items()}, cfg, ids, mask)[:, :-1], ids[:, 1:])
        hid = om.opt_logits({k: v.detach() for k, v in c_sd.items()}, cfg, ids, mask, return_hidden=True)
        scores = (hid @ c_sd['score_head.weight'].detach().t()).squeeze(-1)
        end = torch.stack([m.nonzero()[-1].squeeze() for m in mask])
        o_reward = scores[torch.arange(ids.shape[0]), end]
        o_values = scores[:, :-1]
    # ---------------- native scoring
    d = lambda t: t.to(dev())
    rs = tr.reward_model_step(d(ids), d(mask))
    assert_close(rs['reward'].cpu(), o_reward, rtol=3e-2, atol=3e-2, what='reward (end score at last attended token)')
    sm = mask[:, 1:].bool()
    # position j predicts token j+1: compare where BOTH are attended.  (j = last left-pad position is inside
    # sequence_mask but its query row is fully masked -- HF and the native kernel both return don't-care values
    # there; PPO never uses it because prompt_idx lies beyond the left padding.)
    both = sm & mask[:, :-1].bool()
    assert_close(rs['reward_values'].cpu()[both], o_values[both], rtol=3e-2, atol=3e-2, what='critic values')
    nat_old, _ = tr.sequence_log_probs(tr.actor_reference_model, d(ids), d(mask), 0)
    assert_close(nat_old.cpu()[both], ref_lp[both], rtol=2e-2, atol=5e-2, what='reference log-probs')
    # ---------------- one rl_step on identical rollout statistics (oracle values fed to both sides)
    reward = o_reward * 3.0
    old_values = o_values + 0.3 * torch.randn(o_values.shape, generator=torch.Generator().manual_seed(3))
    tb = {'log_probs': d(old_lp), 'ref_log_probs': d(ref_lp), 'reward': d(reward), 'reward_values': d(old_values), 'prompt_idx': start}
    info = tr.rl_step({'input_ids': d(ids), 'attention_mask': d(mask)}, tb)
    old_rewards = orl.add_kl_divergence_regularization(reward, old_lp, ref_lp, sm, 0.02, 50.0)
    adv, ret = orl.get_advantages_and_returns(old_values, old_rewards, sm, start, 1.0, 0.95)
    lp = orl.gather_log_probabilities(om.opt_logits(a_sd, cfg, ids, mask)[:, :-1], ids[:, 1:])
    a_loss = orl.actor_loss_fn(lp[:, start:], old_lp[:, start:], adv, sm[:, start:], 0.2)
    vals = (om.opt_logits(c_sd, cfg, ids, mask, return_hidden=True) @ c_sd['score_head.weight'].t()).squeeze(-1)[:, :-1]
    c_loss = orl.critic_loss_fn(vals[:, start:], old_values[:, start:], ret, sm[:, start:], 5.0)
    rep = [f"actor_loss native {info['train/actor_loss']:.5f} oracle {a_loss.item():.5f}",
           f"critic_loss native {info['train/reward_critic_loss']:.5f} oracle {c_loss.item():.5f}"]
    assert abs(info['train/actor_loss'] - a_loss.item()) < 3e-2 * max(1.0, abs(a_loss.item()))
    assert abs(info['train/reward_critic_loss'] - c_loss.item()) < 3e-2 * max(1.0, abs(c_loss.item()))
    m = sm[:, start:].float()
    assert abs(info['train/kl_divergence'] - float(((old_lp - ref_lp)[:, start:] * m).sum(-1).mean())) < 1e-3
    assert abs(info['train/reward_with_kl_penalty'] - float((old_rewards[:, start:] * m).sum(-1).mean())) < 1e-3
    assert info['train/max_generated_length'] == float(m.sum(-1).max())
    # gradients the step consumed (still in the flat buffers)
    a_loss.backward(); c_loss.backward()
    for eng, sd, names in ((tr.actor_model, a_sd, ['model.decoder.layers.1.fc1.weight', 'model.decoder.layers.0.self_attn.q_proj.weight',
                                                   'model.decoder.final_layer_norm.weight']),
                           (tr.reward_critic_model, c_sd, ['score_head.weight', 'model.decoder.layers.1.fc2.weight',
                                                           'model.decoder.layers.0.self_attn.v_proj.bias'])):
        eng.wait_optimizer()
        torch.cuda.synchronize()
        for n in names:
            got = eng.module.store.grad_view(n).float().cpu().reshape(sd[n].grad.shape)
            e = rel_err(got, sd[n].grad)
            rep.append(f'{n}: rel_err {e:.4f}')
            assert e < 8e-2, (n, e)
    dump('parity_ppo_rl_step.txt', '\n'.join(rep) + '\n')
    # a second step must run and keep the losses finite (weights were updated by both engines)
    info2 = tr.rl_step({'input_ids': d(ids), 'attention_mask': d(mask)}, tb)
    assert all(map(lambda v: v == v and abs(v) < 1e6, info2.values()))


def test_rm_loss_and_llama_text_model_vs_oracle():
    """Reward-model step (rm.py:97-132) on a GQA Llama backbone: end scores, loss, accuracy and gradients vs the
    oracle (fp32 autograd); also covers NativeLlama (text-only) with num_kv_heads < num_heads."""
    from align_anything_amd import configs
    from align_anything_amd.trainers.rm import RMTrainer
    cfg = configs.llama_cfg(128, 256, 2, 2, 1, 320, rms_eps=1e-5, max_position_embeddings=128)   # GQA: 2 q heads, 1 kv head
    g = torch.Generator().manual_seed(4)
    names = {'model.embed_tokens.weight': (320, 128), 'model.norm.weight': (128,), 'score_head.weight': (1, 128)}
    for i in range(2):
        p = f'model.layers.{i}.'
        names.update({p + 'input_layernorm.weight': (128,), p + 'post_attention_layernorm.weight': (128,),
                      p + 'self_attn.q_proj.weight': (128, 128), p + 'self_attn.k_proj.weight': (64, 128),
                      p + 'self_attn.v_proj.weight': (64, 128), p + 'self_attn.o_proj.weight': (128, 128),
                      p + 'mlp.gate_proj.weight': (256, 128), p + 'mlp.up_proj.weight': (256, 128), p + 'mlp.down_proj.weight': (128, 256)})
    sd = {k: ((torch.randn(s, generator=g) * 0.06) if len(s) == 2 else 1 + 0.1 * torch.randn(s, generator=g)).to(torch.bfloat16) for k, s in names.items()}
    tr = RMTrainer({'train_cfgs': {'regularization': 0.01, 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant',
                                   'weight_decay': 0.0}}, {'gradient_clipping': 1.0}, model_cfg=cfg, state=sd, device='cuda:0')
    N, Tn = 4, 48
    ids = torch.randint(3, 320, (N, Tn), generator=g)
    mask = torch.ones(N, Tn, dtype=torch.long)
    for n, rp in enumerate((0, 9, 20, 3)):   # RIGHT padding
        if rp:
            mask[n, Tn - rp:] = 0; ids[n, Tn - rp:] = 0
    batch = {'input_ids': ids.to(dev()), 'attention_mask': mask.to(dev())}
    ld = tr.loss(batch)
    f = {k: v.float().clone().requires_grad_(True) for k, v in sd.items()}
    x = torch.nn.functional.embedding(ids, f['model.embed_tokens.weight'])
    hid = om.llama_decoder(f, cfg, x, mask.bool(), prefix='model.')
    scores = (hid @ f['score_head.weight'].t()).squeeze(-1)
    end = torch.stack([m.nonzero()[-1].squeeze() for m in mask])
    es = scores[torch.arange(N), end]
    hi, lo = es[:2], es[2:]
    o_loss = -torch.nn.functional.logsigmoid(hi - lo).mean() + 0.01 * torch.stack([lo, hi]).square().mean()
    got_es = torch.cat([ld['higher_end_reward'], ld['lower_end_reward']]).cpu()
    assert_close(got_es, es.detach(), rtol=3e-2, atol=3e-2, what='end scores')
    assert abs(float(ld['loss']) - float(o_loss)) < 2e-2
    assert float(ld['accuracy']) == float((hi > lo).float().mean())
    tr.model.backward(ld['loss'])
    torch.cuda.synchronize()
    o_loss.backward()
    for n in ('score_head.weight', 'model.layers.1.mlp.down_proj.weight', 'model.layers.0.self_attn.k_proj.weight', 'model.norm.weight'):
        got = tr.model.module.store.grad_view(n).float().cpu().reshape(f[n].grad.shape)
        assert rel_err(got, f[n].grad) < 8e-2, (n, rel_err(got, f[n].grad))
    tr.model.step()
    info = tr.train_step(batch)
    assert info['train/loss'] == info['train/loss']


def test_full_ppo_iteration_rollout_then_update():
    """generate -> score -> log-probs -> rl_step, all native (ppo.py:244-289 then 309-398): shapes, masks and
    finiteness; the pieces are each checked against the oracle elsewhere."""
    tr, z, cfg, actor_sd, old_sd, rm_sd, ids, mask, start = _setup()
    tr.cfgs = {'model_cfgs': {'model_max_length': 36, 'temperature': 1.0, 'top_p': 0.95, 'pad_token_id': 1, 'eos_token_id': 2},
               **tr.cfgs}
    prompts = {'input_ids': ids[:, :20].to(dev()), 'attention_mask': mask[:, :20].clone().fill_(1).to(dev())}
    g = torch.Generator(device='cuda').manual_seed(3)
    inf, trn = tr.rollout(prompts, generator=g)
    N, L = inf['input_ids'].shape
    assert L <= 36 and trn['prompt_idx'] == 19
    assert trn['log_probs'].shape == (N, L - 1) == trn['ref_log_probs'].shape == trn['reward_values'].shape
    assert torch.isfinite(trn['reward']).all()
    # actor == reference at this point -> identical log-probs on attended positions
    both = (inf['attention_mask'][:, 1:] & inf['attention_mask'][:, :-1]).bool()
    assert_close(trn['log_probs'][both], trn['ref_log_probs'][both], rtol=0, atol=1e-5, what='actor vs ref log-probs')
    info = tr.rl_step(inf, trn)
    assert all(v == v for v in info.values())
    assert abs(info['train/kl_divergence']) < 1e-4


def test_rl_eval_samples_completions_and_follows_the_reference_schedule():
    """`RLTrainerBase.eval` (base/rl_trainer.py:289-329, common.rl_eval) on the native `generate`: every evaluation prompt gets a sampled
    completion (token ids here: this constructor carries no tokenizer), the evaluation loader is a DevicePrefetcher whose first iterator is
    abandoned (its producer must retire), and `train()` evaluates before the first step and every `eval_interval` steps (ppo.py:422-479)."""
    from align_anything_amd.data import DevicePrefetcher
    tr, z, cfg, actor_sd, old_sd, rm_sd, ids, mask, start = _setup()
    tr.cfgs = {'model_cfgs': {'model_max_length': 30, 'temperature': 1.0, 'top_p': 0.95, 'pad_token_id': 1, 'eos_token_id': 2},
               'data_cfgs': {'eval_datasets': 'handed over as a loader'},
               'train_cfgs': dict(tr.cfgs['train_cfgs'], eval_strategy='steps', eval_interval=2, per_device_train_batch_size=3, epochs=1, update_iters=1)}
    batches = [{'input_ids': ids[:, k:k + 20].contiguous(), 'attention_mask': torch.ones(3, 20, dtype=torch.long)} for k in (0, 10, 20)]
    tr.eval_dataloader = DevicePrefetcher(batches[:2], 'cuda:0')
    first = next(iter(tr.eval_dataloader))
    assert first['input_ids'].is_cuda and torch.equal(first['input_ids'].cpu(), batches[0]['input_ids'])
    ev = tr.eval()
    assert len(ev['eval/prompts']) == 6 and len(ev['eval/generated']) == 6
    assert ev['eval/prompts'][3] == batches[1]['input_ids'][0].tolist()
    assert all(1 <= len(g) <= 10 and all(0 <= t < 320 for t in g) for g in ev['eval/generated'])
    hist = tr.train(DevicePrefetcher(batches, 'cuda:0'), generator=torch.Generator(device='cuda').manual_seed(5))
    assert len(hist) == 3 and [s for s, _ in tr.eval_history] == [0, 2] and all(v == v for v in hist[-1].values())


def test_qwen2_style_text_model_dpo_step_vs_oracle():
    """Llama block with q/k/v biases and GQA (Qwen2 family, from_hf_config 'qwen2'): DPO loss and gradients
    (incl. the fused bias gradient) against the oracle."""
    from align_anything_amd import configs
    from align_anything_amd.trainers.dpo import DPOTrainer
    cfg = configs.llama_cfg(128, 256, 2, 2, 1, 320, rms_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=128, attention_bias=True)
    g = torch.Generator().manual_seed(9)
    names = {'model.embed_tokens.weight': (320, 128), 'model.norm.weight': (128,), 'lm_head.weight': (320, 128)}
    for i in range(2):
        p = f'model.layers.{i}.'
        names.update({p + 'input_layernorm.weight': (128,), p + 'post_attention_layernorm.weight': (128,),
                      p + 'self_attn.q_proj.weight': (128, 128), p + 'self_attn.k_proj.weight': (64, 128),
                      p + 'self_attn.v_proj.weight': (64, 128), p + 'self_attn.o_proj.weight': (128, 128),
                      p + 'self_attn.q_proj.bias': (128,), p + 'self_attn.k_proj.bias': (64,), p + 'self_attn.v_proj.bias': (64,),
                      p + 'mlp.gate_proj.weight': (256, 128), p + 'mlp.up_proj.weight': (256, 128), p + 'mlp.down_proj.weight': (128, 256)})
    def init(k, s):
        if k.endswith('norm.weight'):
            return 1 + 0.1 * torch.randn(s, generator=g)
        return torch.randn(s, generator=g) * (0.3 if k.endswith('bias') else 0.06)
    sd = {k: init(k, s).to(torch.bfloat16) for k, s in names.items()}
    sdr = {k: (v.float() + 0.02 * torch.randn(v.shape, generator=g)).to(torch.bfloat16) for k, v in sd.items()}
    tr = DPOTrainer({'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant'},
                     'model_cfgs': {'pad_token_id': 0}}, {'gradient_clipping': 1.0}, model_cfg=cfg, policy_state=sd, reference_state=sdr, device='cuda:0')
    N, Tn = 4, 48
    ids = torch.randint(3, 320, (N, Tn), generator=g); mask = torch.ones(N, Tn, dtype=torch.long)
    for n, lp in enumerate((0, 7, 3, 0)):
        ids[n, :lp] = 0; mask[n, :lp] = 0
    lens = [9, 12, 5, 16]
    batch = {'input_ids': ids.to(dev()), 'attention_mask': mask.to(dev()), 'meta_info': {'response_lens': lens}}
    ld = tr.loss(batch)
    f = {k: v.float().clone().requires_grad_(True) for k, v in sd.items()}
    fr = {k: v.float() for k, v in sdr.items()}
    lp = orl.compute_log_probs(om.llama_logits(f, cfg, ids, mask), ids, lens, 0)
    with torch.no_grad():
        rlp = orl.compute_log_probs(om.llama_logits(fr, cfg, ids, mask), ids, lens, 0)
    o = orl.dpo_loss(lp, rlp, 0.1)
    assert abs(float(ld['loss']) - float(o['loss'])) < 1e-2
    tr.model.backward(ld['loss']); torch.cuda.synchronize()
    o['loss'].backward()
    for n in ('model.layers.0.self_attn.q_proj.bias', 'model.layers.1.self_attn.v_proj.bias', 'model.layers.1.self_attn.k_proj.weight',
              'model.layers.0.mlp.up_proj.weight', 'lm_head.weight'):
        got = tr.policy.store.grad_view(n).float().cpu().reshape(f[n].grad.shape)
        assert rel_err(got, f[n].grad) < 8e-2, (n, rel_err(got, f[n].grad))


def test_rule_reward_replaces_the_reward_model():
    """trainers/text_to_text/ppo_remote_rm.py:321-347: the reward comes from a scorer outside the model (rule / HTTP service); the critic
    values and the whole update are unchanged.  A scalar return is promoted to 1-D as the reference does (:337-338)."""
    from align_anything_amd.trainers.ppo import PPOTrainer
    tr, z, cfg, actor_sd, old_sd, rm_sd, ids, mask, start = _setup()
    want = tr.reward_model_step(ids.to(dev()), mask.to(dev()))
    seen = {}

    def rule(i, a):
        seen['shape'] = tuple(i.shape)
        return [float(x) for x in (i * a).sum(1) % 7]            # a host-side list, like remote_rm_client.score's return

    tr2 = PPOTrainer(tr.cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, actor_state=actor_sd, critic_state=rm_sd, device='cuda:0', reward_fn=rule)
    assert tr2.reward_model is None
    got = tr2.reward_model_step(ids.to(dev()), mask.to(dev()))
    assert seen['shape'] == tuple(ids.shape)
    assert torch.equal(got['reward'].cpu(), ((ids * mask).sum(1) % 7).float())
    assert torch.equal(got['reward_values'], want['reward_values'])          # same critic weights -> identical values
    lp, _ = tr2.sequence_log_probs(tr2.actor_model, ids.to(dev()), mask.to(dev()))
    info = tr2.rl_step({'input_ids': ids.to(dev()), 'attention_mask': mask.to(dev())},
                       {'prompt_idx': start, 'log_probs': lp, 'ref_log_probs': lp.clone(), 'reward': got['reward'], 'reward_values': got['reward_values']})
    assert abs(info['train/reward'] - float(got['reward'].mean())) < 1e-6 and info['train/kl_divergence'] == 0.0
    bad = PPOTrainer(tr.cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, actor_state=actor_sd, critic_state=rm_sd, device='cuda:0',
                     reward_fn=lambda i, a: torch.zeros(2))
    with pytest.raises(ValueError):
        bad.reward_model_step(ids.to(dev()), mask.to(dev()))


# ====================================================================== pinned to the REFERENCE's own trainers (tests/golden/opt_tiny_rm.npz,
# opt_tiny_ppo.npz: oracle/gen_golden.py::gen_opt_rm / gen_opt_ppo drive the unmodified RMTrainer.loss and PPOTrainer.rollout / rl_step)
@pytest.mark.parametrize('dtype', ['bf16', 'fp32'])
def test_rm_trainer_loss_matches_reference_fixture(dtype):
    """All six outputs of the reference's RMTrainer.loss (trainers/text_to_text/rm.py:97-132) on its AccustomedOPTRewardModel,
    right-padded batch, with and without regularisation; gradients of the backward it would run."""
    from align_anything_amd.trainers.rm import RMTrainer
    z = load_golden('opt_tiny_rm.npz')
    ids, mask = T(z['input_ids']).to(dev()), T(z['attention_mask']).to(dev())
    f32 = dtype == 'fp32'
    rep = []
    for tag in ('reg', 'noreg'):
        sd = state_dict_from_golden(z, 'w.', torch.float32 if f32 else torch.bfloat16)
        tr = RMTrainer({'train_cfgs': {'regularization': float(z[f'{tag}_regularization']), 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0,
                                       'lr_scheduler_type': 'constant', 'weight_decay': 0.0, 'compute_dtype': dtype}},
                       {'gradient_clipping': 1.0}, model_cfg=tiny_opt_cfg(), state=sd, device='cuda:0')
        ld = tr.loss({'input_ids': ids, 'attention_mask': mask})
        assert set(ld) >= {'loss', 'higher_end_reward', 'lower_end_reward', 'higher_rewards', 'lower_rewards', 'accuracy'}
        rms = float(T(z[f'{tag}_higher_rewards']).pow(2).mean().sqrt())      # the scores of this fixture have rms ~5: bf16 error scales with it
        tol = dict(rtol=1e-4, atol=1e-4) if f32 else dict(rtol=3e-2, atol=3e-2 * max(1.0, rms))
        for k in ('higher_end_reward', 'lower_end_reward'):
            assert_close(ld[k].cpu(), T(z[f'{tag}_{k}']), what=f'{tag} {k}', **tol)
        valid = T(z['attention_mask']).bool()
        for k, rows in (('higher_rewards', slice(0, 3)), ('lower_rewards', slice(3, 6))):
            assert ld[k].shape == (3, ids.shape[1])
            assert_close(ld[k].cpu()[valid[rows]], T(z[f'{tag}_{k}'])[valid[rows]], what=f'{tag} {k}', **tol)
        assert abs(float(ld['loss']) - float(z[f'{tag}_loss'])) < (2e-5 if f32 else 6e-2), (float(ld['loss']), float(z[f'{tag}_loss']))
        assert float(ld['accuracy']) == float(z[f'{tag}_accuracy'])
        tr.model.backward(ld['loss'])
        torch.cuda.synchronize()
        worst = 0.0
        for k in z.files:
            if k.startswith(f'{tag}_g.'):
                name = k[len(tag) + 3:]
                got = tr.model.module.store.grad_view(name).float().cpu().reshape(z[k].shape)
                worst = max(worst, rel_err(got, T(z[k])))
        rep.append(f'{dtype} {tag}: loss {float(ld["loss"]):.6f} vs reference {float(z[f"{tag}_loss"]):.6f}, worst gradient rel-err {worst:.2e}')
        assert worst < (2e-5 if f32 else 1.2e-1), rep          # bf16: the envelope of profiles/parity/parity_bf16_envelope.txt on a 2-layer model with x3 weights
    dump(f'parity_rm_reference_{dtype}.txt', '\n'.join(rep) + '\n')


def test_vlm_reward_models_take_the_score_at_the_last_position():
    """models/llava.py:64-68 / qwen2_vl.py:61-64: end_scores = score of position -1 whatever the mask says; OPT / Llama take the
    last ATTENDED token (models/opt.py:67).  On a right-padded RM batch the two differ, and the RM trainer must follow the backbone."""
    from align_anything_amd.trainers.common import end_index
    from align_anything_amd.trainers.rm import RMTrainer
    from tests.util import tiny_llava_cfg
    mask = torch.ones(4, 40, dtype=torch.long)
    mask[1, 30:] = 0
    mask[3, 12:] = 0
    assert end_index('opt', mask).tolist() == [39, 29, 39, 11] and end_index('qwen3moe', mask).tolist() == [39, 29, 39, 11]
    assert end_index('llava', mask).tolist() == [39] * 4 and end_index('qwen2vl', mask).tolist() == [39] * 4
    z = load_golden('llava_tiny_dpo.npz')
    sd = {k: v for k, v in state_dict_from_golden(z, 'w.', torch.bfloat16).items() if k != 'lm_head.weight'}
    g = torch.Generator().manual_seed(5)
    sd['score_head.weight'] = (torch.randn(1, 128, generator=g) * 0.3).to(torch.bfloat16)
    tr = RMTrainer({'train_cfgs': {'regularization': 0.0, 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant'}},
                   {'gradient_clipping': 1.0}, model_cfg=tiny_llava_cfg(), state=sd, device='cuda:0')
    ids, am = T(z['input_ids']).clone(), T(z['attention_mask']).clone()
    am[1, -6:] = 0                                                                # right padding on one row (ids stay: the tower needs its image tokens)
    b = {'input_ids': ids.to(dev()), 'attention_mask': am.to(dev()), 'pixel_values': T(z['pixel_values']).to(dev())}
    ld = tr.loss(b)
    scores = torch.cat([ld['higher_rewards'], ld['lower_rewards']])
    end = torch.cat([ld['higher_end_reward'], ld['lower_end_reward']])
    assert_close(end.cpu(), scores[:, -1].cpu(), rtol=0, atol=1e-6, what='VLM end score = score at position -1')
    tr.model.backward(ld['loss'])
    torch.cuda.synchronize()
    assert torch.isfinite(tr.model.module.store.grad_view('score_head.weight').float()).all()


@pytest.mark.parametrize('dtype', ['bf16', 'fp32'])
def test_llava_rm_trainer_loss_matches_reference_fixture(dtype):
    """The reference's text+image RMTrainer.loss (trainers/text_image_to_text/rm.py -> text_to_text/rm.py:97-132) on its
    AccustomedLlavaRewardModel (models/llava.py:47-76), tests/golden/llava_tiny_rm.npz: six outputs + gradients on the collator's
    left-padded batch and with one row's mask cut on the RIGHT -- there the reference takes the score at position -1, a masked position."""
    from align_anything_amd.trainers.rm import RMTrainer
    from tests.test_oracle_golden import _llava_rm_state
    from tests.util import tiny_llava_cfg
    z = load_golden('llava_tiny_rm.npz')
    f32 = dtype == 'fp32'
    ids, pix = T(z['input_ids']).to(dev()), T(z['pixel_values']).to(dev())
    B = ids.shape[0] // 2
    rep = []
    for tag in ('left', 'rightcut'):
        mask = T(z[f'{tag}_attention_mask']).to(dev())
        sd = _llava_rm_state(z, torch.float32 if f32 else torch.bfloat16)
        tr = RMTrainer({'train_cfgs': {'regularization': float(z['regularization']), 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0,
                                       'lr_scheduler_type': 'constant', 'weight_decay': 0.0, 'compute_dtype': dtype}},
                       {'gradient_clipping': 1.0}, model_cfg=tiny_llava_cfg(), state=sd, device='cuda:0')
        ld = tr.loss({'input_ids': ids, 'attention_mask': mask, 'pixel_values': pix})
        rms = float(T(z[f'{tag}_higher_rewards']).pow(2).mean().sqrt())
        # bf16: per-position scores of a 2-layer model with x3 weights carry ~4 % of their rms in rounding noise (one of 93 read 0.079 off
        # at rms 1.86 in the first hardware run); the fp32 twin pins the same numbers to 1e-4
        tol = dict(rtol=1e-4, atol=1e-4) if f32 else dict(rtol=3e-2, atol=6e-2 * max(1.0, rms))
        for k in ('higher_end_reward', 'lower_end_reward'):
            assert_close(ld[k].cpu(), T(z[f'{tag}_{k}']), what=f'{tag} {k}', **tol)
        valid = T(z[f'{tag}_attention_mask']).bool()
        for k, rows in (('higher_rewards', slice(0, B)), ('lower_rewards', slice(B, 2 * B))):
            assert_close(ld[k].cpu()[valid[rows]], T(z[f'{tag}_{k}'])[valid[rows]], what=f'{tag} {k}', **tol)
        assert abs(float(ld['loss']) - float(z[f'{tag}_loss'])) < (2e-5 if f32 else 6e-2), (tag, float(ld['loss']), float(z[f'{tag}_loss']))
        assert float(ld['accuracy']) == float(z[f'{tag}_accuracy'])
        tr.model.backward(ld['loss'])
        torch.cuda.synchronize()
        worst, n = 0.0, 0
        for k in z.files:
            if k.startswith(f'{tag}_g.'):
                name = k[len(tag) + 3:]
                name = name if name == 'score_head.weight' else name[len('model.'):]
                g = tr.model.module.store.grad_view(name)
                if g is None:                    # the CLIP tower is frozen by default natively (configs/train/text_image_to_text/*.yaml)
                    assert 'vision_tower' in name, name
                    continue
                worst = max(worst, rel_err(g.float().cpu().reshape(z[k].shape), T(z[k])))
                n += 1
        assert n == 5 and worst < (2e-5 if f32 else 1.2e-1), (tag, n, worst)
        rep.append(f'{dtype} {tag}: loss {float(ld["loss"]):.6f} vs reference {float(z[f"{tag}_loss"]):.6f}, worst gradient rel-err {worst:.2e} over {n} tensors')
    dump(f'parity_llava_rm_reference_{dtype}.txt', '\n'.join(rep) + '\n')


@pytest.mark.parametrize('dtype', ['bf16', 'fp32'])
def test_t2t_ppo_rollout_and_rl_step_match_reference_fixture(dtype):
    """The reference's own text_to_text PPOTrainer.rollout (fixed `generate` output) and rl_step, micro-batch by micro-batch:
    log-probs / reference log-probs / end reward / critic values of the rollout, then the 12 metrics and the gradients of rl_step."""
    from align_anything_amd.trainers.ppo import PPOTrainer
    z = load_golden('opt_tiny_ppo.npz')
    f32 = dtype == 'fp32'
    wd = torch.float32 if f32 else torch.bfloat16
    a_sd, r_sd = state_dict_from_golden(z, 'a.', wd), state_dict_from_golden(z, 'r.', wd)
    rm_sd, c_sd = state_dict_from_golden(z, 'rm.', wd), state_dict_from_golden(z, 'c.', wd)
    cfgs = {'train_cfgs': {'actor_lr': 1e-3, 'critic_lr': 1e-3, 'actor_weight_decay': 0.0, 'critic_weight_decay': 0.0, 'actor_lr_warmup_ratio': 0.0,
                           'critic_lr_warmup_ratio': 0.0, 'actor_lr_scheduler_type': 'constant', 'critic_lr_scheduler_type': 'constant',
                           'kl_coeff': float(z['kl_coeff']), 'clip_range_ratio': float(z['clip_range_ratio']), 'clip_range_value': float(z['clip_range_value']),
                           'clip_range_score': float(z['clip_range_score']), 'gamma': float(z['gamma']), 'gae_lambda': float(z['gae_lambda']),
                           'per_device_train_batch_size': int(z['micro']), 'compute_dtype': dtype},
            'model_cfgs': {'pad_token_id': int(z['pad_token_id']), 'eos_token_id': int(z['eos_token_id'])}}
    micro, P = int(z['micro']), z['prompts'].shape[1]
    rep = []
    for i in range(z['sequences'].shape[0] // micro):
        # a fresh trainer per micro-batch: the fixture's stand-in engines never stepped, so every rl_step starts from the same weights
        tr = PPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=tiny_opt_cfg(), actor_state=a_sd, reward_state=rm_sd, critic_state=c_sd, device='cuda:0')
        tr.actor_reference_model.module.load_state_dict(r_sd)
        rows = slice(i * micro, (i + 1) * micro)
        prompts = T(z['prompts'])[rows].to(dev())
        pb = {'input_ids': prompts, 'attention_mask': prompts.ne(int(z['pad_token_id'])).long()}
        inf, trn = tr.rollout(pb, sequences=T(z['sequences'])[rows].to(dev()))
        assert torch.equal(inf['input_ids'].cpu(), T(z[f'mb{i}.input_ids'])) and trn['prompt_idx'] == int(z[f'mb{i}.prompt_idx']) == P - 1
        assert torch.equal(inf['attention_mask'].cpu(), T(z[f'mb{i}.attention_mask']))
        am = T(z[f'mb{i}.attention_mask'])
        both = (am[:, 1:] & am[:, :-1]).bool()
        tol = dict(rtol=1e-4, atol=2e-4) if f32 else dict(rtol=3e-2, atol=6e-2)
        for k in ('log_probs', 'ref_log_probs', 'reward_values'):
            assert_close(trn[k].cpu()[both], T(z[f'mb{i}.{k}'])[both], what=f'mb{i} {k}', **tol)
        assert_close(trn['reward'].cpu(), T(z[f'mb{i}.reward']), what=f'mb{i} reward', **tol)
        # rl_step on the FIXTURE's experience (the reference's numbers in, its metrics and gradients out)
        tb = {k: T(z[f'mb{i}.{k}']).to(dev()) for k in ('log_probs', 'ref_log_probs', 'reward', 'reward_values')}
        tb['prompt_idx'] = P - 1
        info = tr.rl_step(inf, tb)
        lim = 2e-5 if f32 else 3e-2
        for k in ('train/actor_loss', 'train/reward_critic_loss', 'train/reward', 'train/reward_with_kl_penalty', 'train/reward_advantage',
                  'train/reward_return', 'train/reward_value', 'train/kl_divergence', 'train/mean_generated_length', 'train/max_generated_length'):
            want = float(z[f'mb{i}.info.{k}'])
            assert abs(info[k] - want) < lim * max(1.0, abs(want)), (i, k, info[k], want)
            rep.append(f'{dtype} mb{i} {k}: native {info[k]:.6f} reference {want:.6f}')
        worst = 0.0
        for eng, tag in ((tr.actor_model, f'mb{i}.ga.'), (tr.reward_critic_model, f'mb{i}.gc.')):
            eng.wait_optimizer()
            torch.cuda.synchronize()
            for k in z.files:
                if k.startswith(tag):
                    got = eng.module.store.grad_view(k[len(tag):]).float().cpu().reshape(z[k].shape)
                    worst = max(worst, rel_err(got, T(z[k])))
        rep.append(f'{dtype} mb{i} worst gradient rel-err {worst:.2e}')
        assert worst < (3e-5 if f32 else 1.2e-1), rep
    dump(f'parity_ppo_t2t_reference_{dtype}.txt', '\n'.join(rep) + '\n')
