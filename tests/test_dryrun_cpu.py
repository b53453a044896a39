"""CPU: host control flow of every trainer with the kernel launches stubbed out.

The product path has no CPU fallback, so nothing here computes: `ops.call` (the single ctypes gateway) is replaced by a recorder,
the device checks of ops.py are relaxed, and the trainers run on CPU tensors whose contents are whatever `torch.empty` holds.  What
this pins without a GPU is everything python does between the launches -- constructor wiring, batch / window plumbing, shapes of
every buffer handed to a kernel, engine bookkeeping (accumulation, schedules, optimizer ordering), metric keys -- for the code
paths the GPU suite covers AND for the ones written after the round's GPU budget was spent (PPOTrainer.ptx_step).  Numerical
parity lives in the `-m gpu` tests."""
import numpy as np
import pytest
import torch

from tests.util import load_golden, state_dict_from_golden, tiny_llava_cfg, tiny_opt_cfg, tiny_qwen3moe_cfg

T = torch.from_numpy


@pytest.fixture
def launches(monkeypatch):
    from align_anything_amd import ops
    seen = []

    def sfx(t, name):
        if t.dtype == torch.bfloat16:
            return ''
        if t.dtype == torch.float32:
            return '_f32'
        raise RuntimeError(f'{name}: expected bfloat16 or float32 activations, got {t.dtype}')

    monkeypatch.setattr(ops, 'call', lambda name, *a: seen.append(name))
    monkeypatch.setenv('AA_VALIDATE_FIRST_BATCH', '0')      # the first-batch checks read kernel OUTPUTS (the image-token count), which do not exist here
    monkeypatch.setattr(ops, '_sfx', sfx)
    monkeypatch.setattr(ops, '_chk', lambda t, dtype, name: None)
    monkeypatch.setattr(ops, 'stream', lambda: 0)
    return seen


def _cfgs(z=None, **extra):
    t = {'scale_coeff': 0.1, 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant', 'weight_decay': 0.0}
    t.update(extra)
    return {'train_cfgs': t, 'model_cfgs': {'pad_token_id': int(z['pad_token_id']) if z is not None else 1}}


def _pref_batch(z, pixels=False):
    b = {'input_ids': T(z['input_ids']), 'attention_mask': T(z['attention_mask']), 'meta_info': {'response_lens': [int(x) for x in z['response_lens']]}}
    if pixels:
        b['pixel_values'] = T(z['pixel_values'])
    return b


@pytest.mark.parametrize('dtype', ['bf16', 'fp32'])
def test_dpo_family_steps_launch_the_expected_kernels(launches, dtype):
    from align_anything_amd.trainers.dpo import DPOTrainer
    from align_anything_amd.trainers.pref import KTOTrainer, ORPOTrainer, SimPOTrainer
    z = load_golden('opt_tiny_dpo.npz')
    wd = torch.bfloat16 if dtype == 'bf16' else torch.float32
    keys = {'train/loss', 'train/reward', 'train/better_sample_reward', 'train/worse_sample_reward', 'train/reward_accuracy', 'train/reward_margin', 'train/lr'}
    for cls, loss_kernel in ((DPOTrainer, 'aa_dpo_loss_fwd_bwd'), (SimPOTrainer, 'aa_pref_loss_fwd_bwd'), (ORPOTrainer, 'aa_pref_loss_fwd_bwd'), (KTOTrainer, 'aa_pref_loss_fwd_bwd')):
        tr = cls(_cfgs(z, compute_dtype=dtype, gradient_accumulation_steps=2), {'gradient_clipping': 1.0}, model_cfg=tiny_opt_cfg(),
                 policy_state=state_dict_from_golden(z, 'w.', wd), reference_state=state_dict_from_golden(z, 'r.', wd), device='cpu')
        assert (tr.reference is None) == (cls in (SimPOTrainer, ORPOTrainer))
        del launches[:]
        info = tr.train_step(_pref_batch(z))
        assert keys <= set(info) and loss_kernel in launches
        assert 'aa_adamw_flat' not in launches and tr.model.global_steps == 0         # accumulation: no optimizer step at the first micro-batch
        tr.train_step(_pref_batch(z))
        assert launches.count('aa_adamw_flat') == len(tr.policy.store.trainable_groups()) and tr.model.global_steps == 1
        assert 'aa_grad_sumsq' in launches and 'aa_clip_coef' in launches
        gemm = 'aa_gemm_bf16' if dtype == 'bf16' else 'aa_gemm_f32'
        assert gemm in launches and ('aa_attn_bwd' + ('' if dtype == 'bf16' else '_f32')) in launches
        tr.train_dataloader = [_pref_batch(z)] * 4                 # the inherited epoch loop (accumulation of 2 -> 2 optimizer steps)
        del launches[:]
        hist = tr.train([_pref_batch(z)] * 3) if cls is KTOTrainer else tr.train()
        assert len(hist) == 4 and hist[-1]['train/epoch'] == 1.0 and tr.model.global_steps == 3
        assert launches.count('aa_window_kl') == (3 if cls is KTOTrainer else 0)


def test_llava_and_moe_dpo_steps(launches):
    from align_anything_amd.trainers.dpo import DPOTrainer
    z = load_golden('llava_tiny_dpo.npz')
    tr = DPOTrainer(_cfgs(z), {'gradient_clipping': 1.0}, model_cfg=tiny_llava_cfg(), policy_state=state_dict_from_golden(z, 'w.', torch.bfloat16),
                    reference_state=state_dict_from_golden(z, 'r.', torch.bfloat16), device='cpu')
    info = tr.train_step(_pref_batch(z, pixels=True))
    assert np.isfinite(info['train/lr'])
    for k in ('aa_patch_im2col', 'aa_clip_embed', 'aa_image_slot_index', 'aa_embed_fwd', 'aa_layernorm_fwd', 'aa_rmsnorm_bwd', 'aa_dpo_loss_fwd_bwd'):
        assert k in launches, k
    z = load_golden('qwen3moe_tiny_dpo.npz')
    del launches[:]
    tr = DPOTrainer(_cfgs(z), {'gradient_clipping': 1.0}, model_cfg=tiny_qwen3moe_cfg(), policy_state=state_dict_from_golden(z, 'w.', torch.bfloat16),
                    reference_state=state_dict_from_golden(z, 'r.', torch.bfloat16), device='cpu')
    tr.train_step(_pref_batch(z))
    for k in ('aa_moe_route', 'aa_moe_plan', 'aa_moe_gather', 'aa_gemm_grouped_bf16', 'aa_moe_combine', 'aa_moe_combine_bwd', 'aa_moe_route_bwd'):
        assert k in launches, k


def test_llava_dpo_step_with_shared_prompt_packing_host_flow(launches):
    """train_cfgs.share_prompt_prefix: the host flow of a packed DPO step on CPU tensors (launches recorded): the plan is built from the batch, every row-wise
    kernel is handed the PACKED row count, attention the reference layout between two row gathers, and the backward gathers / sums the copies."""
    from align_anything_amd import ops
    from align_anything_amd.trainers.dpo import DPOTrainer
    z = load_golden('llava_tiny_dpo.npz')
    g = torch.Generator().manual_seed(0)
    B, Tn, P, Rc, Rr = 2, 192, (100, 120), (40, 30), (64, 30)
    ids = torch.full((2 * B, Tn), 301, dtype=torch.long)
    mask = torch.zeros_like(ids)
    for i in range(B):
        prompt = torch.cat([torch.tensor([1]), torch.full((4,), 300), torch.randint(3, 299, (P[i] - 5,), generator=g)])
        for row, R in ((i, Rc[i]), (B + i, Rr[i])):
            seq = torch.cat([prompt, torch.randint(3, 299, (R,), generator=g)])
            ids[row, Tn - len(seq):] = seq
            mask[row, Tn - len(seq):] = 1
    pix = torch.randn(B, 3, 28, 28, generator=g)
    batch = {'input_ids': ids, 'attention_mask': mask, 'pixel_values': torch.cat([pix, pix], 0), 'meta_info': {'response_lens': list(Rc) + list(Rr)}}
    tr = DPOTrainer(_cfgs(z, share_prompt_prefix=True), {'gradient_clipping': 1.0}, model_cfg=tiny_llava_cfg(), policy_state=state_dict_from_golden(z, 'w.', torch.bfloat16),
                    reference_state=state_dict_from_golden(z, 'r.', torch.bfloat16), device='cpu')
    rows = []
    real_gather = ops.moe_gather
    ops_gemm = ops.gemm

    def gemm(a, b, out=None, **kw):
        rows.append(a.shape[1] if kw.get('a_t') else a.shape[0])
        return ops_gemm(a, b, out=out, **kw)
    ops.gemm = gemm
    try:
        info = tr.train_step(batch)
    finally:
        ops.gemm = ops_gemm
    assert real_gather is ops.moe_gather and np.isfinite(info['train/lr'])
    plan = batch['_pack']
    tokens = int(mask.sum())
    assert plan is not None and plan['prefix_lens'] == [100, 120] and plan['rows'] == tokens - 220 and plan['Mq'] % 64 == 0
    # per forward (policy + reference): layer 0 gathers qkv and the attention output, the LAST layer qkv only (its attention output goes straight to the window
    # rows: dead-row elimination); the backward gathers d_attn and the residual gradient back in the last layer, d_attn in layer 0, and sums the copies of d_qkv per layer
    assert launches.count('aa_moe_gather') == 2 * 3 + 3 and launches.count('aa_gather2_add') == 2
    assert plan['Mq'] in rows and 2 * B * Tn not in rows          # the projections ran on the packed rows, none on the reference layout
    assert 'aa_attn_fwd_qskip' in launches and 'aa_attn_bwd_qskip' in launches and 'aa_dpo_loss_fwd_bwd' in launches      # attention leaves out the query blocks nobody consumes
    # off by default: the same batch without the switch takes the reference layout
    tr0 = DPOTrainer(_cfgs(z), {'gradient_clipping': 1.0}, model_cfg=tiny_llava_cfg(), policy_state=state_dict_from_golden(z, 'w.', torch.bfloat16),
                     reference_state=state_dict_from_golden(z, 'r.', torch.bfloat16), device='cpu')
    b0 = {k: v for k, v in batch.items() if not k.startswith('_')}
    del launches[:]
    tr0.train_step(b0)
    # ... and there the only gathers are the two of the last layer's dead-row elimination (d_attn and the residual gradient back to the [N, T] layout), with the
    # attention of that layer leaving out the queries nobody consumes (policy + reference forward, policy backward)
    assert b0.get('_pack') is None and launches.count('aa_moe_gather') == 2 and 'aa_gather2_add' not in launches
    assert launches.count('aa_attn_fwd_qskip') == 2 and launches.count('aa_attn_bwd_qskip') == 1


def test_supervised_step_and_prefetched_window(launches):
    from align_anything_amd.data import DevicePrefetcher
    from align_anything_amd.trainers.sft import SupervisedTrainer
    z, zw = load_golden('opt_tiny_sft.npz'), load_golden('opt_tiny_dpo.npz')
    tr = SupervisedTrainer(_cfgs(zw), {'gradient_clipping': 1.0}, model_cfg=tiny_opt_cfg(), policy_state=state_dict_from_golden(zw, 'w.', torch.bfloat16), device='cpu')
    b = {'input_ids': T(z['input_ids']), 'attention_mask': T(z['attention_mask']), 'labels': T(z['labels'])}
    info = tr.train_step(b)
    assert set(info) == {'train/loss', 'train/lr'} and 'aa_sft_loss_fwd_bwd' in launches and 'aa_adamw_flat' in launches
    (pb,) = list(DevicePrefetcher([b], 'cpu'))
    assert pb['_window']['rows'] == int((T(z['labels'])[:, 1:] != -100).sum())
    tr.train_step(pb)                                      # the pre-built plan is taken as is
    assert tr.model.global_steps == 2


def test_ppo_rollout_update_and_ptx_steps(launches, tmp_path):
    from align_anything_amd.trainers.ppo import PPOTrainer
    z = load_golden('opt_tiny_dpo.npz')
    cfg = tiny_opt_cfg()
    actor_sd = state_dict_from_golden(z, 'w.', torch.bfloat16)
    rm_sd = {k: v for k, v in actor_sd.items() if k != 'lm_head.weight'}
    rm_sd['score_head.weight'] = torch.zeros(1, cfg['hidden_size'], dtype=torch.bfloat16)
    cfgs = {'train_cfgs': {'actor_lr': 1e-3, 'critic_lr': 1e-3, 'actor_lr_scheduler_type': 'constant', 'critic_lr_scheduler_type': 'constant', 'ptx_coeff': 4.0},
            'model_cfgs': {'pad_token_id': 1, 'model_max_length': 30, 'temperature': 0.9, 'top_p': 0.8, 'repetition_penalty': 1.2}}
    tr = PPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, actor_state=actor_sd, reward_state=rm_sd, device='cpu')
    assert tr.ptx_coeff == 4.0
    prompts = T(z['input_ids'])[:, :24]
    inference, training = tr.rollout({'input_ids': prompts, 'attention_mask': T(z['attention_mask'])[:, :24]})
    assert inference['input_ids'].shape == (4, 30) and torch.equal(inference['input_ids'][:, :24], prompts)
    assert training['log_probs'].shape == training['ref_log_probs'].shape == (4, 29) and training['prompt_idx'] == 23
    for k in ('aa_gemm_skinny_bf16', 'aa_attn_decode', 'aa_sample_top_k_top_p', 'aa_mark_seen', 'aa_rowdot_fwd'):
        assert k in launches, k
    info = tr.rl_step(inference, training)
    assert {'train/actor_loss', 'train/reward_critic_loss', 'train/kl_divergence', 'train/actor_lr'} <= set(info)
    assert tr.actor_model.global_steps == 1 and tr.reward_critic_model.global_steps == 1
    labels = inference['input_ids'].clone()
    labels[:, :24] = -100
    del launches[:]
    out = tr.ptx_step({'input_ids': inference['input_ids'], 'attention_mask': inference['attention_mask'], 'labels': labels})
    assert set(out) == {'train/ptx_loss'} and tr.actor_model.global_steps == 2 and tr.reward_critic_model.global_steps == 1
    assert 'aa_sft_loss_fwd_bwd' in launches and 'aa_adamw_flat' in launches
    # the outer loop: 2 prompt batches x 2 micro-batches x 2 update iterations, PTX batches cycled and split into one-row micro-batches
    cfgs2 = {'train_cfgs': dict(cfgs['train_cfgs'], per_device_train_batch_size=2, update_iters=2, epochs=1), 'model_cfgs': cfgs['model_cfgs']}
    tr3 = PPOTrainer(cfgs2, {'gradient_clipping': 1.0}, model_cfg=cfg, actor_state=actor_sd, reward_state=rm_sd, device='cpu')
    pbatch = {'input_ids': prompts, 'attention_mask': T(z['attention_mask'])[:, :24]}
    ptx = {'input_ids': inference['input_ids'], 'attention_mask': inference['attention_mask'], 'labels': labels}
    hist = tr3.train([pbatch, pbatch], ptx_dataloader=[ptx])
    assert len(hist) == 8 and tr3.global_step == 8 and all('train/ptx_loss' in h and 'train/actor_loss' in h for h in hist)
    # base/rl_trainer.py:231-234: with PTX the actor accumulates over the (rl_step, ptx_step) pair -> ONE update per pair
    assert tr3.actor_model.gas == 2 and tr3.actor_model.global_steps == 8 and tr3.reward_critic_model.global_steps == 8
    assert tr3.actor_model.micro_steps == 16
    with pytest.raises(RuntimeError):
        tr3.train([pbatch])                                # dropping the PTX set mid-run would change the accumulation depth
    tr4 = PPOTrainer(cfgs2, {'gradient_clipping': 1.0}, model_cfg=cfg, actor_state=actor_sd, reward_state=rm_sd, device='cpu')
    assert len(tr4.train([pbatch])) == 4 and tr4.actor_model.gas == 1 and tr4.actor_model.global_steps == 4          # without PTX
    # the default schedulers (cosine actor, constant critic) get their length from the prompt dataloader like the reference:
    # 2 prompt batches x 1 epoch x 2 update_iters x per_device_train_batch_size 2 x per_device_prompt_batch_size 1 = 8 micro steps
    cfgs3 = {'train_cfgs': {k: v for k, v in cfgs2['train_cfgs'].items() if not k.endswith('scheduler_type')}, 'model_cfgs': cfgs['model_cfgs']}
    cfgs3['train_cfgs'].update(actor_lr_warmup_ratio=0.0, critic_lr_warmup_ratio=0.0)
    tr5 = PPOTrainer(cfgs3, {'gradient_clipping': 1.0}, model_cfg=cfg, actor_state=actor_sd, reward_state=rm_sd, device='cpu')
    assert tr5.actor_model.sched == 'cosine' and tr5.actor_model.total_steps is None
    with pytest.raises(RuntimeError, match='cosine'):
        tr5.rl_step(inference, training)                   # no schedule length known: refuse instead of decaying to 0
    tr5 = PPOTrainer(cfgs3, {'gradient_clipping': 1.0}, model_cfg=cfg, actor_state=actor_sd, reward_state=rm_sd, device='cpu')
    hist = tr5.train([pbatch, pbatch])
    assert tr5.actor_model.total_steps == 8 and tr5.reward_critic_model.total_steps == 8
    lrs = [h['train/actor_lr'] for h in hist]
    assert len(lrs) == 8 and all(a > b for a, b in zip(lrs, lrs[1:])) and lrs[1] > 0.5e-3 and lrs[-1] == 0.0, lrs
    # an actor / critic RESUMED through load_checkpoint under the default cosine schedule has steps behind it but no schedule length; train() must
    # still hand it one (the RM / DPO / GRPO fix, here too) -- and a second train() on the same trainer keeps the schedule it started with
    tr5.actor_model.save_checkpoint(str(tmp_path / 'ppo_actor'))
    tr5.reward_critic_model.save_checkpoint(str(tmp_path / 'ppo_critic'))
    tr6 = PPOTrainer(cfgs3, {'gradient_clipping': 1.0}, model_cfg=cfg, actor_state=actor_sd, reward_state=rm_sd, device='cpu')
    tr6.actor_model.load_checkpoint(str(tmp_path / 'ppo_actor'))
    tr6.reward_critic_model.load_checkpoint(str(tmp_path / 'ppo_critic'))
    assert tr6.actor_model.global_steps == 8 and tr6.actor_model.total_steps is None
    tr6._set_schedules([pbatch, pbatch, pbatch], False)
    assert tr6.actor_model.total_steps == 12 and tr6.actor_model.global_steps == 8 and tr6.actor_model.micro_steps == 8
    from align_anything_amd.engine import cosine_with_warmup
    assert abs(tr6.actor_model.optimizer.param_groups[0]['lr'] - cosine_with_warmup(8, tr6.actor_model.base_lr, 0, 12)) < 1e-15
    tr6._set_schedules([pbatch], False)                    # already scheduled and stepped: kept
    assert tr6.actor_model.total_steps == 12
    # ... and a slice saved at one accumulation depth, loaded at another, restarts at the boundary of its last completed update
    e6 = tr6.actor_model
    e6.gas, e6.micro_steps = 4, 8 * 4 + 3
    e6.save_checkpoint(str(tmp_path / 'ppo_actor_gas4'))
    e6.gas = 2
    e6.load_checkpoint(str(tmp_path / 'ppo_actor_gas4'))
    assert e6.micro_steps == 16 and e6.global_steps == 8 and e6.micro_steps % e6.gas == 0
    e6.gas = 1
    # a rule reward instead of the reward model
    tr2 = PPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, actor_state=actor_sd, critic_state=rm_sd, device='cpu', reward_fn=lambda i, a: [1.0] * i.shape[0])
    assert tr2.reward_model is None and tr2.reward_model_step(prompts, torch.ones_like(prompts))['reward'].tolist() == [1.0] * 4


def test_grpo_and_rm_steps(launches):
    from align_anything_amd.trainers.grpo import GRPOTrainer
    from align_anything_amd.trainers.rm import RMTrainer
    z = load_golden('opt_tiny_dpo.npz')
    cfg = tiny_opt_cfg()
    sd = state_dict_from_golden(z, 'w.', torch.bfloat16)
    cfgs = {'train_cfgs': {'actor_lr': 1e-3, 'actor_lr_scheduler_type': 'constant', 'beta': 0.04, 'num_generations': 2},
            'model_cfgs': {'pad_token_id': 1, 'eos_token_id': 2, 'model_max_length': 28, 'temperature': 1.0, 'top_p': 1.0}}
    tr = GRPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, actor_state=sd, reference_state=sd, reward_fn=lambda c: c.sum(1).float(), device='cpu')
    prompts = T(z['input_ids'])[:2, :20]
    info = tr.train_step({'input_ids': prompts, 'attention_mask': torch.ones_like(prompts)})
    assert {'train/loss', 'train/reward'} <= set(info)
    for k in ('aa_group_advantage', 'aa_completion_mask', 'aa_grpo_loss_fwd_bwd', 'aa_adamw_flat'):
        assert k in launches, k
    rm_sd = {k: v for k, v in sd.items() if k != 'lm_head.weight'}
    rm_sd['score_head.weight'] = torch.zeros(1, cfg['hidden_size'], dtype=torch.bfloat16)
    rm = RMTrainer(_cfgs(z), {'gradient_clipping': 1.0}, model_cfg=cfg, state=rm_sd, device='cpu')
    del launches[:]
    info = rm.train_step(_pref_batch(z))
    assert 'train/loss' in info and 'aa_rm_loss_fwd_bwd' in launches and 'aa_rowdot_bwd' in launches
    assert len(rm.train([_pref_batch(z)] * 3)) == 3 and rm.model.global_steps == 4 and rm.global_step == 3
    assert len(tr.train([{'input_ids': prompts, 'attention_mask': torch.ones_like(prompts)}] * 2)) == 2 and tr.actor_model.global_steps == 3
    assert rm.eval() == {} and rm.eval([]) == {}
    del launches[:]
    ev = rm.eval([_pref_batch(z), _pref_batch(z)])
    assert set(ev) == {'eval/accuracy', 'eval/reward_mean', 'eval/reward_std'} and launches.count('aa_rowdot_fwd') == 2 and 'aa_adamw_flat' not in launches


def test_multimodal_backbones_and_ti2t_ppo_update(launches):
    from align_anything_amd.trainers.dpo import DPOTrainer
    from align_anything_amd.trainers.ppo_ti2t import PPOTrainerTI2T
    from tests.util import bits_to_bf16, tiny_qwen2audio_cfg, tiny_qwen2vl_cfg
    # Qwen2-VL DPO (vision tower -> merger -> multimodal RoPE tables -> decoder), with the visual blocks training
    z = load_golden('qwen2vl_tiny_dpo.npz')
    tr = DPOTrainer(_cfgs(z, freeze_vision_tower=False), {'gradient_clipping': 1.0}, model_cfg=tiny_qwen2vl_cfg(), policy_state=state_dict_from_golden(z, 'w.', torch.bfloat16),
                    reference_state=state_dict_from_golden(z, 'r.', torch.bfloat16), device='cpu')
    b = _pref_batch(z, pixels=True)
    b['image_grid_thw'] = T(z['image_grid_thw'])
    info = tr.train_step(b)
    assert 'train/loss' in info
    for k in ('aa_mrope_tables', 'aa_rope_inplace', 'aa_attn_bwd', 'aa_adamw_flat'):
        assert k in launches, k
    # Qwen2-Audio DPO (conv front-end as im2col GEMMs, trainable tower, identical-pair skipping mask)
    z = load_golden('qwen2audio_tiny_dpo.npz')
    sd = lambda pre: {k[len(pre):]: (bits_to_bf16(z[k]) if z[k].dtype == np.uint16 else T(z[k]).to(torch.bfloat16)) for k in z.files if k.startswith(pre)}
    pol, ref = sd('w.'), sd('r.')
    ref.setdefault('model.audio_tower.embed_positions.weight', pol['model.audio_tower.embed_positions.weight'])
    del launches[:]
    tr = DPOTrainer(_cfgs(z), {'gradient_clipping': 1.0}, model_cfg=tiny_qwen2audio_cfg(), policy_state=pol, reference_state=ref, device='cpu')
    assert tr.skip_identical_pairs
    b = _pref_batch(z)
    b['input_features'], b['feature_attention_mask'] = T(z['input_features']), T(z['feature_attention_mask'])
    tr.train_step(b)
    for k in ('aa_conv1d_im2col', 'aa_conv1d_col2im', 'aa_avgpool2', 'aa_attn_fwd'):
        assert k in launches, k
    # ti2t PPO update on the reference's own rollout statistics (tail windows, response mask)
    z = load_golden('qwen2vl_tiny_ppo.npz')
    cfgs = {'train_cfgs': {'actor_lr': 1e-3, 'critic_lr': 1e-3, 'actor_lr_scheduler_type': 'constant', 'critic_lr_scheduler_type': 'constant'},
            'model_cfgs': {'pad_token_id': int(z['pad_token_id']), 'max_new_tokens': 10, 'eos_token_id': 2}}
    wd = torch.bfloat16
    vis = {k: v for k, v in state_dict_from_golden(z, 'a.', wd).items() if k.startswith('model.visual.')}
    score_sd = lambda tag: {**vis, **{k: v for k, v in state_dict_from_golden(z, tag + '.', wd).items() if k != 'lm_head.weight'}}
    tr = PPOTrainerTI2T(cfgs, {'gradient_clipping': 1.0}, model_cfg=tiny_qwen2vl_cfg(), actor_state=state_dict_from_golden(z, 'a.', wd),
                        reward_state=score_sd('rm'), critic_state=score_sd('c'), device='cpu')
    inf = {'input_ids': T(z['sequences_left']), 'attention_mask': T(z['attention_mask']), 'pixel_values': T(z['pixel_values']), 'image_grid_thw': T(z['image_grid_thw'])}
    trn = {k: (T(z[k]) if k != 'response_lens' else z[k].tolist()) for k in ('response_lens', 'log_probs', 'ref_log_probs', 'reward', 'reward_values', 'response_mask')}
    del launches[:]
    info = tr.rl_step(inf, trn)
    assert {'train/actor_loss', 'train/reward_critic_loss', 'train/mean_generated_length', 'train/max_generated_length'} <= set(info)
    assert info['train/max_generated_length'] == float(T(z['response_mask']).sum(-1).max())      # host arithmetic on real inputs
    for k in ('aa_kl_reward', 'aa_gae', 'aa_ppo_actor_loss', 'aa_ppo_critic_loss'):
        assert k in launches, k


def test_rollout_plumbing_on_llama_and_moe_decoders(launches):
    """generate(): prefill with KV sink, strip-major weight copies (Llama family), per-row expert GEMV (Qwen3-MoE), token bookkeeping."""
    from align_anything_amd.generation import generate
    from align_anything_amd.modeling import build_model
    z = load_golden('llava_tiny_dpo.npz')
    m = build_model(tiny_llava_cfg(), 'cpu', trainable=False)
    m.load_state_dict(state_dict_from_golden(z, 'w.', torch.bfloat16))
    ids, mask = T(z['input_ids'])[:, :30], T(z['attention_mask'])[:, :30]
    seq = generate(m, ids, mask, max_new_tokens=5, do_sample=False, pad_token_id=301, pixel_values=T(z['pixel_values']))
    assert seq.shape == (ids.shape[0], 35) and torch.equal(seq[:, :30], ids)
    assert 'aa_swizzle_weights_bf16' in launches and 'aa_gemm_skinny_swz_bf16' in launches and 'aa_decode_rope_cache' in launches and 'aa_argmax_rows' in launches
    n_swz = launches.count('aa_swizzle_weights_bf16')
    generate(m, ids, mask, max_new_tokens=2, do_sample=False, pad_token_id=301, pixel_values=T(z['pixel_values']))
    assert launches.count('aa_swizzle_weights_bf16') == 2 * n_swz            # refreshed in place every call, same storage
    assert m.stack._dw is not None
    m.stack.release_decode()
    assert m.stack._dw is None
    z = load_golden('qwen3moe_tiny_dpo.npz')
    m = build_model(tiny_qwen3moe_cfg(), 'cpu', trainable=False)
    m.load_state_dict(state_dict_from_golden(z, 'w.', torch.bfloat16))
    del launches[:]
    ids, mask = T(z['input_ids'])[:, :20], T(z['attention_mask'])[:, :20]
    seq = generate(m, ids, mask, max_new_tokens=4, do_sample=True, temperature=0.7, top_p=0.9, pad_token_id=int(z['pad_token_id']))
    assert seq.shape == (4, 24) and 'aa_moe_gemv_bf16' in launches and 'aa_sample_top_k_top_p' in launches
    del launches[:]
    generate(m, ids.repeat(5, 1), mask.repeat(5, 1), max_new_tokens=2, do_sample=False, pad_token_id=int(z['pad_token_id']))     # 20 rows: tile layout
    assert 'aa_gemm_grouped_bf16' in launches and 'aa_moe_gemv_bf16' not in launches


def _dp_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from align_anything_amd import ops
    from align_anything_amd.trainers.dpo import DPOTrainer
    seen = []
    ops.call = lambda name, *a: seen.append(name)
    ops._sfx = lambda t, name: '' if t.dtype == torch.bfloat16 else '_f32'
    ops._chk = lambda t, dtype, name: None
    ops.stream = lambda: 0
    z = load_golden('opt_tiny_dpo.npz')
    tr = DPOTrainer(_cfgs(z), {'gradient_clipping': 1.0}, model_cfg=tiny_opt_cfg(), policy_state=state_dict_from_golden(z, 'w.', torch.bfloat16),
                    reference_state=state_dict_from_golden(z, 'r.', torch.bfloat16), device='cpu')
    st = tr.policy.store
    ok = tr.model.world == world
    # every gradient element must pass through exactly one all-reduce: mark the buffers with the rank's id and look at the sums
    orig_zero = st.zero_grad
    def zero_then_mark():
        orig_zero()
        for g in st.gflat.values():
            g.fill_(float(rank + 1))
    st.zero_grad = zero_then_mark
    rows = [rank, rank + 2]
    b = {'input_ids': T(z['input_ids'])[rows], 'attention_mask': T(z['attention_mask'])[rows], 'meta_info': {'response_lens': [int(z['response_lens'][r]) for r in rows]}}
    info = tr.train_step(b)
    want = float(sum(range(1, world + 1)))
    for g in st.gflat.values():
        ok = ok and bool((g.float() == want).all())
    ok = ok and 'train/loss' in info and tr.model.global_steps == 1 and 'aa_adamw_flat' in seen
    # checkpoints: every rank calls save(), rank 0 alone writes (replicas are identical)
    import tempfile
    d = tempfile.mkdtemp(prefix=f'aa_dp_save_r{rank}_')
    tr.save(tag=1, output_dir=d)
    wrote = os.path.exists(os.path.join(d, 'slice_1', 'pytorch_model.bin'))
    ok = ok and wrote == (rank == 0)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_reduces_every_gradient_element_exactly_once():
    """World-size-2 gloo run of a whole DPO train_step with stubbed launches: the per-layer buckets issued during backward plus the
    remainder buckets cover every element of every gradient buffer once (sum of the ranks' marks), metrics are reduced, one optimizer step."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def _ep_worker(rank, world, port, q):
    """Expert-parallel DPO step, stubbed launches -- except that the two INTEGER kernels whose outputs steer the host (router choice,
    expert-major plan) are emulated by their definitions, written straight into the (host) buffers the C ABI would have filled."""
    import ctypes
    import os
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from align_anything_amd import ops
    from align_anything_amd.trainers.dpo import DPOTrainer
    seen = []
    ints = lambda ptr, n: np.ctypeslib.as_array((ctypes.c_int * n).from_address(ptr))

    def call(name, *a):
        seen.append(name)
        if name.startswith('aa_moe_route') and 'bwd' not in name:          # (logits, ld, rows, E, k, norm, probs, idx, weights, stream)
            rows, E, k, idx = a[2], a[3], a[4], a[7]
            g = np.random.default_rng(1000 * rank + len(seen))
            ints(idx, rows * k)[:] = np.stack([g.permutation(E)[:k] for _ in range(rows)]).reshape(-1)
        elif name == 'aa_moe_plan':                                         # the plan of csrc/moe.hip, restated (stable sort by expert)
            idx, rows, k, E, align, cap, counts, off, pos, src, te = a[:11]
            flat = ints(idx, rows * k).copy()
            valid = (flat >= 0) & (flat < E)               # -1 = a row of the capacity-padded exchange without a token: in no segment, pos untouched
            cnt = np.bincount(flat[valid], minlength=E)
            seg = (cnt + align - 1) // align * align
            o = np.concatenate([[0], np.cumsum(seg)])
            ints(counts, E)[:] = cnt
            ints(off, E + 1)[:] = o
            order = np.argsort(np.where(valid, flat, E), kind='stable')[:int(valid.sum())]
            starts = np.cumsum(cnt) - cnt
            dest = o[flat[order]] + (np.arange(order.size) - starts[flat[order]])
            if order.size:
                ints(pos, rows * k)[order] = dest
            s_arr = ints(src, cap)
            s_arr[:] = -1
            s_arr[dest] = order // k
            tg = 128 if align % 128 == 0 else align       # rows per table entry (csrc/moe.hip)
            t_arr = ints(te, cap // tg)
            t_arr[:] = -1
            for e in range(E):
                t_arr[o[e] // tg:o[e + 1] // tg] = e

    ops.call = call
    ops._sfx = lambda t, name: '' if t.dtype == torch.bfloat16 else '_f32'
    ops._chk = lambda t, dtype, name: None
    ops.stream = lambda: 0
    z = load_golden('qwen3moe_tiny_dpo.npz')
    tr = DPOTrainer(_cfgs(z, expert_parallel=True), {'gradient_clipping': 1.0}, model_cfg=tiny_qwen3moe_cfg(), policy_state=state_dict_from_golden(z, 'w.', torch.bfloat16),
                    reference_state=state_dict_from_golden(z, 'r.', torch.bfloat16), device='cpu')
    st = tr.policy.store
    ok = tr.policy.ep.size == world and st.p['model.layers.0.mlp.experts.gate_up_proj'].shape[0] == 8 // world and 'exp' in st.sizes
    orig_zero = st.zero_grad

    def zero_then_mark():
        orig_zero()
        for g in st.gflat.values():
            g.fill_(float(rank + 1))
    st.zero_grad = zero_then_mark
    rows = [rank, rank + 2]
    b = {'input_ids': T(z['input_ids'])[rows], 'attention_mask': T(z['attention_mask'])[rows], 'meta_info': {'response_lens': [int(z['response_lens'][r]) for r in rows]}}
    info = tr.train_step(b)
    want = float(sum(range(1, world + 1)))
    for name, g in st.gflat.items():       # replicated groups are summed over the ranks; the expert shard is this rank's own and never reduced
        ok = ok and bool((g.float() == (float(rank + 1) if name == 'exp' else want)).all())
    ok = ok and 'train/loss' in info and 'aa_gemm_grouped_bf16' in seen and seen.count('aa_moe_plan') == 2 * 2 * 2      # (dense + local plan) x 2 layers x (policy + reference)
    sd = tr.policy.state_dict()                                            # collective: expert rows gathered back
    ok = ok and sd['model.layers.1.mlp.experts.down_proj'].shape[0] == 8
    # rollout on the sharded weights: one token exchange per decode position, so the ranks must run the SAME number of passes although
    # their prompts (hence their own budgets under one max_length) differ: 24 - 10 = 14 new tokens on rank 0, 24 - 16 = 8 on rank 1
    from align_anything_amd.generation import generate
    L = 10 if rank == 0 else 16
    ids, mask = T(z['input_ids'])[rows, :L], T(z['attention_mask'])[rows, :L]
    del seen[:]
    seq = generate(tr.policy, ids, mask, max_length=24, do_sample=False, pad_token_id=int(z['pad_token_id']))
    passes = seen.count('aa_attn_decode')
    ok = ok and seq.shape == (2, 24) and passes == 13 * 2                  # 14 positions -> 13 decode passes x 2 layers, on BOTH ranks
    del seen[:]
    generate(tr.policy, ids, mask, max_length=24, do_sample=False, eos_token_id=7, sync_every=2, pad_token_id=int(z['pad_token_id']))
    both = [None] * world
    dist.all_gather_object(both, seen.count('aa_attn_decode'))              # whatever the (garbage) tokens did, the ranks stopped together
    ok = ok and len(set(both)) == 1
    # a rank with no room for new tokens (prompt length == max_length) must not leave the others in the exchange: EVERY rank raises
    L2 = 10 if rank == 0 else 24
    try:
        generate(tr.policy, T(z['input_ids'])[rows, :L2], T(z['attention_mask'])[rows, :L2], max_length=24, do_sample=False, pad_token_id=int(z['pad_token_id']))
        raised = False
    except ValueError:
        raised = True
    ok = ok and raised
    # the RL trainers' flag: a PPO trainer with train_cfgs.expert_parallel shards actor / reference / reward / critic (one communicator),
    # rolls out in lockstep and takes one rl_step (actor + critic updates; expert shards never all-reduced)
    from align_anything_amd.trainers.ppo import PPOTrainer
    wsd = state_dict_from_golden(z, 'w.', torch.bfloat16)
    rsd = {k: v for k, v in wsd.items() if k != 'lm_head.weight'}
    rsd['score_head.weight'] = torch.zeros(1, tiny_qwen3moe_cfg()['hidden_size'], dtype=torch.bfloat16)
    pcfgs = {'train_cfgs': {'expert_parallel': True, 'actor_lr': 1e-3, 'critic_lr': 1e-3, 'actor_lr_scheduler_type': 'constant',
                            'critic_lr_scheduler_type': 'constant'},
             'model_cfgs': {'pad_token_id': int(z['pad_token_id']), 'model_max_length': 24}}
    ptr = PPOTrainer(pcfgs, {'gradient_clipping': 1.0}, model_cfg=tiny_qwen3moe_cfg(), actor_state=wsd, reward_state=rsd, device='cpu')
    eps = [m.module.ep for m in (ptr.actor_model, ptr.actor_reference_model, ptr.reward_model, ptr.reward_critic_model)]
    ok = ok and all(e is not None and e.size == world for e in eps) and len({id(e) for e in eps}) == 1
    inference, training = ptr.rollout({'input_ids': ids, 'attention_mask': mask})
    ok = ok and inference['input_ids'].shape == (2, 24)
    info = ptr.rl_step(inference, training)
    ok = ok and 'train/actor_loss' in info and ptr.actor_model.global_steps == 1 and ptr.reward_critic_model.global_steps == 1
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_expert_parallel_step_host_flow():
    """World-size-2 gloo run of an expert-parallel Qwen3-MoE DPO step on CPU tensors: the token exchange runs for real (split sizes from
    the emulated plan), the replicated gradient groups are all-reduced exactly once, the expert shard never is, the clip-norm exchange and the
    optimizer step go through, `state_dict()` gathers the experts.  The numerics of the same step are pinned on hardware (tests/test_ep_gpu.py)."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ep_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


@pytest.mark.skipif(not __import__('os').path.isdir('/root/reference/align_anything'),
                    reason='the reference checkout exists only in the build container (never on the GPU box)')
def test_integration_level_b_stub_runs_the_reference_train_step(launches, monkeypatch):
    """INTEGRATION.md section 3, executed: the three overrides a maintainer adds to the REFERENCE's own DPOTrainer
    (init_models / init_engines / compute_log_probs+loss) -- then the reference's UNMODIFIED train_step
    (align_anything/trainers/text_to_text/dpo.py:205-237) drives the native engines: loss dict keys, engine.backward(loss),
    engine.step(), optimizer.param_groups[0]['lr'].  Kernel launches are recorded, not executed (CPU)."""
    from oracle import _shim
    _shim.install()
    import align_anything.trainers.text_to_text.dpo as ref_dpo
    from transformers import OPTConfig, OPTForCausalLM
    from align_anything_amd import configs
    from align_anything_amd.engine import NativeEngine
    from align_anything_amd.trainers.dpo import DPOTrainer as NativeDPO

    z = load_golden('opt_tiny_dpo.npz')
    oc = OPTConfig(hidden_size=128, ffn_dim=256, num_hidden_layers=2, num_attention_heads=2, vocab_size=320, max_position_embeddings=128,
                   word_embed_proj_dim=128, dropout=0.0, attention_dropout=0.0, pad_token_id=1)
    hf = OPTForCausalLM(oc)

    def hf_load(self):                                     # stands in for load_pretrained_models (no checkpoint / network here)
        self.model, self.tokenizer, self.processor = hf, type('Tok', (), {'pad_token_id': 1})(), None
        self.reference_model = hf
    monkeypatch.setattr(ref_dpo.DPOTrainer, 'init_models', hf_load)
    monkeypatch.setattr(ref_dpo, 'get_all_reduce_mean', lambda t: t)            # world size 1, no process group

    class DPOTrainer(ref_dpo.DPOTrainer):                  # === the stub of INTEGRATION.md section 3, verbatim ===
        def init_models(self):
            super().init_models()                          # HF load: tokenizer/processor + checkpoint tensors
            cfg = configs.from_hf_config(self.model.config)
            sd = self.model.state_dict()
            self.native = NativeDPO(self.cfgs, self.ds_train_cfgs, model_cfg=cfg, policy_state=sd, reference_state=sd,
                                    tokenizer=self.tokenizer, train_dataloader=self.train_dataloader, device='cpu')
            del self.model, self.reference_model           # the HF modules are not used on the hot path

        def init_engines(self):
            self.model, self.reference_model = self.native.model, self.native.reference_model

        compute_log_probs = lambda self, model, batch: self.native.compute_log_probs(model, batch)
        loss = lambda self, batch: self.native.loss(batch)

    tr = DPOTrainer.__new__(DPOTrainer)                    # the reference __init__ wants yaml / datasets / deepspeed configs
    tr.cfgs, tr.ds_train_cfgs = _cfgs(z), {'gradient_clipping': 1.0}
    batch = _pref_batch(z)
    tr.train_dataloader = [batch, batch]
    tr.init_models()
    tr.init_engines()
    assert isinstance(tr.model, NativeEngine) and isinstance(tr.reference_model, NativeEngine) and not hasattr(tr, 'native_model')
    del launches[:]
    info = ref_dpo.DPOTrainer.train_step(tr, batch)        # the reference's own method body
    assert set(info) == {'train/loss', 'train/reward', 'train/better_sample_reward', 'train/worse_sample_reward',
                         'train/reward_accuracy', 'train/reward_margin', 'train/lr'}
    assert info['train/lr'] == 1e-3 and tr.model.global_steps == 1
    for k in ('aa_gemm_bf16', 'aa_attn_fwd', 'aa_dpo_loss_fwd_bwd', 'aa_attn_bwd', 'aa_grad_sumsq', 'aa_adamw_flat'):
        assert k in launches, k
    # log-probs: the fused lm_head x log-prob walk (default) or the unfused GEMM + gather pair (AA_LMHEAD_FUSED=0)
    assert 'aa_lmhead_logprob_fwd' in launches or 'aa_logprob_gather_fwd' in launches
    lp = ref_dpo.DPOTrainer.compute_log_probs is not DPOTrainer.compute_log_probs and tr.compute_log_probs(tr.model, batch)
    assert lp.shape == (4, max(int(r) for r in z['response_lens']) - 1)
    # the reference's checkpoint call on the engine (supervised_trainer.py:404-450) works on the native engine too
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        tr.model.save_16bit_model(d, save_filename='pytorch_model.bin')
        sd = torch.load(d + '/pytorch_model.bin')
        assert set(sd) >= set(hf.state_dict()) - {'lm_head.weight'}


def test_train_resumes_mid_epoch_and_saves_on_the_reference_schedule(launches, tmp_path):
    """dpo.py:256-270 (remaining epochs, the first `global_step % len(dataloader)` batches skipped) and dpo.py:285-293
    (`slice_<global_step>` every epochs * len(dataloader) // save_total_limit steps)."""
    import os
    from align_anything_amd.trainers.dpo import DPOTrainer
    z = load_golden('opt_tiny_dpo.npz')
    cfgs = _cfgs(z, epochs=2)
    cfgs['logger_cfgs'] = {'output_dir': str(tmp_path), 'save_total_limit': 3}
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=tiny_opt_cfg(), policy_state=state_dict_from_golden(z, 'w.', torch.bfloat16),
                    reference_state=state_dict_from_golden(z, 'r.', torch.bfloat16), device='cpu')
    tr.train_dataloader = [_pref_batch(z)] * 3
    tr.global_step = 4                                     # a checkpoint taken after step 4 of 6: one batch of epoch 2 is already consumed
    hist = tr.train()
    assert len(hist) == 2 and tr.global_step == 6 and hist[-1]['train/epoch'] == 2.0
    # save interval = 2 * 3 // 3 = 2 -> only step 6 falls in the resumed part
    assert sorted(os.listdir(tmp_path)) == ['slice_6'] and os.path.exists(tmp_path / 'slice_6' / 'pytorch_model.bin')
    tr.global_step = 0
    hist = tr.train()
    assert len(hist) == 6 and sorted(os.listdir(tmp_path)) == ['slice_2', 'slice_4', 'slice_6']


def test_rm_grpo_ppo_loops_resume_and_save_on_the_reference_schedules(launches, tmp_path):
    """rm.py:276-314 (resume + slice every epochs * len // save_total_limit), grpo.py:347-386 (the same + the final save of the actor),
    ppo.py:462-468 (slice every total_update_steps // save_total_limit) -- the sibling loops of dpo.py:256-293."""
    import os
    from align_anything_amd.trainers.grpo import GRPOTrainer
    from align_anything_amd.trainers.ppo import PPOTrainer
    from align_anything_amd.trainers.rm import RMTrainer
    z = load_golden('opt_tiny_dpo.npz')
    cfg = tiny_opt_cfg()
    sd = state_dict_from_golden(z, 'w.', torch.bfloat16)
    rm_sd = {k: v for k, v in sd.items() if k != 'lm_head.weight'}
    rm_sd['score_head.weight'] = torch.zeros(1, cfg['hidden_size'], dtype=torch.bfloat16)
    # ---- RM: 2 epochs x 3 batches, limit 3 -> every 2 steps; resumed after step 4 (one batch of epoch 2 consumed)
    c = _cfgs(z, epochs=2)
    c['logger_cfgs'] = {'output_dir': str(tmp_path / 'rm'), 'save_total_limit': 3}
    rm = RMTrainer(c, {'gradient_clipping': 1.0}, model_cfg=cfg, state=rm_sd, device='cpu')
    rm.global_step = 4
    assert len(rm.train([_pref_batch(z)] * 3)) == 2 and rm.global_step == 6
    assert sorted(os.listdir(tmp_path / 'rm')) == ['slice_6'] and os.path.exists(tmp_path / 'rm' / 'slice_6' / 'pytorch_model.bin')
    saved = torch.load(tmp_path / 'rm' / 'slice_6' / 'pytorch_model.bin')
    assert 'score_head.weight' in saved and 'lm_head.weight' not in saved
    # ---- an RM engine resumed by load_checkpoint under the DEFAULT cosine schedule: the constructor cannot know the schedule length (the
    # dataloader is handed to train()), load_checkpoint sets global_steps > 0 -- train() must still give the schedule its length instead of raising
    c2 = _cfgs(z, epochs=2)
    c2['train_cfgs'].update(lr_scheduler_type='cosine', lr_warmup_ratio=0.0, learning_rate=1e-3)
    rm2 = RMTrainer(c2, {'gradient_clipping': 1.0}, model_cfg=cfg, state=rm_sd, device='cpu')
    assert rm2.model.total_steps is None
    rm2.model.save_checkpoint(str(tmp_path / 'eng'))
    rm2.model.global_steps = 99
    rm2.model.load_checkpoint(str(tmp_path / 'eng'))
    assert rm2.model.global_steps == 0 and rm2.model.micro_steps == 0
    # a slice taken INSIDE an accumulation window: micro_steps travels with it, and the gradient buffers of the loading engine are cleared (the window's
    # earlier micro-batches are lost, as with DeepSpeed; nothing stale may be accumulated into)
    rm2.model.gas, rm2.model.micro_steps, rm2.model.global_steps = 2, 3, 1
    rm2.model.save_checkpoint(str(tmp_path / 'eng2'))
    for g in rm2.model.module.store.gflat.values():
        g.fill_(1.0)
    rm2.model.micro_steps = 0
    rm2.model.load_checkpoint(str(tmp_path / 'eng2'))
    assert rm2.model.micro_steps == 3 and rm2.model.global_steps == 1 and all(float(g.float().abs().sum()) == 0.0 for g in rm2.model.module.store.gflat.values())
    rm2.model.gas = 1
    rm2.model.global_steps, rm2.model.micro_steps, rm2.global_step = 4, 4, 4          # = a slice taken after step 4 of 6
    hist2 = rm2.train([_pref_batch(z)] * 3)
    from align_anything_amd.engine import cosine_with_warmup
    assert len(hist2) == 2 and rm2.model.total_steps == 6 and abs(hist2[0]['train/lr'] - cosine_with_warmup(5, 1e-3, 0, 6)) < 1e-12
    # ---- GRPO: 1 epoch x 4 prompt batches, limit 2 -> slices at 2 and 4, and the final actor (output_dir is configured)
    prompts = T(z['input_ids'])[:2, :20]
    pb = {'input_ids': prompts, 'attention_mask': torch.ones_like(prompts)}
    gc = {'train_cfgs': {'actor_lr': 1e-3, 'actor_lr_scheduler_type': 'constant', 'num_generations': 2},
          'model_cfgs': {'pad_token_id': 1, 'eos_token_id': 2, 'model_max_length': 24},
          'logger_cfgs': {'output_dir': str(tmp_path / 'grpo'), 'save_total_limit': 2}}
    g = GRPOTrainer(gc, {'gradient_clipping': 1.0}, model_cfg=cfg, actor_state=sd, reference_state=sd, reward_fn=lambda x: x.sum(1).float(), device='cpu')
    assert len(g.train([pb] * 4)) == 4
    assert sorted(os.listdir(tmp_path / 'grpo')) == ['slice_2', 'slice_4', 'slice_end']
    assert 'lm_head.weight' in torch.load(tmp_path / 'grpo' / 'slice_end' / 'pytorch_model.bin') or cfg['kind'] == 'opt'
    g.global_step = 3                                     # resumed: only the last prompt batch is left
    assert len(g.train([pb] * 4)) == 1 and g.global_step == 4
    # ---- PPO: 2 prompt batches x 2 micro-batches x 1 update iteration = total_update_steps 2 * 1 * 1 * 2 * 1 = 4, limit 2 -> slices at 2 and 4
    pc = {'train_cfgs': {'actor_lr': 1e-3, 'critic_lr': 1e-3, 'actor_lr_scheduler_type': 'constant', 'critic_lr_scheduler_type': 'constant',
                         'per_device_train_batch_size': 2, 'update_iters': 1, 'epochs': 1},
          'model_cfgs': {'pad_token_id': 1, 'model_max_length': 28},
          'logger_cfgs': {'output_dir': str(tmp_path / 'ppo'), 'save_total_limit': 2}}
    p = PPOTrainer(pc, {'gradient_clipping': 1.0}, model_cfg=cfg, actor_state=sd, reward_state=rm_sd, device='cpu')
    four = {'input_ids': T(z['input_ids'])[:, :24], 'attention_mask': T(z['attention_mask'])[:, :24]}
    assert len(p.train([four, four])) == 4 and p.global_step == 4
    assert sorted(os.listdir(tmp_path / 'ppo')) == ['slice_2', 'slice_4']


def test_dpo_trainer_builds_itself_from_cfgs_like_the_reference(launches, tmp_path):
    """VERDICT r3 missing #2 / #3: `DPOTrainer(cfgs, ds_cfgs)` alone, as the reference's constructor (text_to_text/dpo.py:59-77): the models come
    from `model_cfgs.model_name_or_path` (a sharded HF directory, checkpoint.load_pretrained), the dataloader from `data_cfgs` through the
    reference's OWN PreferenceDataset / ChatTemplate / PreferenceCollator (init_datasets -> common.get_dataloaders) on the reference's own
    asset file, and `train()` runs the epoch.  Kernel launches are recorded, not executed (CPU)."""
    import os
    ref_assets = '/root/reference/assets/text_to_text/preference/train.json'
    if not os.path.exists(ref_assets):
        pytest.skip('the reference package (dataset / template plugins) is only present in the build container')
    from oracle import _shim
    _shim.install()
    import transformers as tf
    from tokenizers import Tokenizer, models, pre_tokenizers
    from align_anything_amd.data import DevicePrefetcher
    from align_anything_amd.trainers.dpo import DPOTrainer
    vocab = {w: i for i, w in enumerate(['<s>', '</s>', '<unk>'] + [f'w{i}' for i in range(317)])}
    tk = Tokenizer(models.WordLevel(vocab, unk_token='<unk>'))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    fast = tf.PreTrainedTokenizerFast(tokenizer_object=tk, bos_token='<s>', eos_token='</s>', unk_token='<unk>')
    fast.chat_template = "{% for m in messages %}{{ m['role'] }} : {{ m['content'] }} </s> {% endfor %}"
    torch.manual_seed(0)
    hf = tf.OPTForCausalLM(tf.OPTConfig(hidden_size=128, ffn_dim=256, num_hidden_layers=2, num_attention_heads=2, vocab_size=320, max_position_embeddings=600,
                                        word_embed_proj_dim=128, dropout=0.0, pad_token_id=1)).eval()
    d = str(tmp_path / 'opt')
    hf.save_pretrained(d, max_shard_size='100KB')
    fast.save_pretrained(d)
    cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant', 'weight_decay': 0.0,
                           'per_device_train_batch_size': 4, 'epochs': 1},
            'model_cfgs': {'model_name_or_path': d, 'model_max_length': 512},
            'data_cfgs': {'train_datasets': ref_assets, 'train_template': 'PKUSafeRLHF', 'train_size': None, 'train_split': None, 'train_name': None,
                          'train_data_files': None, 'train_optional_args': []}}
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, device='cpu')
    # models: geometry from config.json (+1 vocab row: the tokenizer had no pad token), weights streamed from the shards, tokenizer from the directory
    assert tr.model_cfg['kind'] == 'opt' and tr.model_cfg['vocab_size'] == 321 and tr.pad_token_id == 320 and tr.tokenizer.padding_side == 'left'
    emb = tr.policy.state_dict()['model.decoder.embed_tokens.weight']
    want = hf.state_dict()['model.decoder.embed_tokens.weight']
    assert torch.equal(emb[:320].float(), want.to(torch.bfloat16).float()) and emb.shape[0] == 321
    assert torch.equal(tr.reference.state_dict()['model.decoder.layers.1.fc2.weight'].float(),
                       hf.state_dict()['model.decoder.layers.1.fc2.weight'].to(torch.bfloat16).float())
    # datasets: the reference's dataset over its 32-pair asset, batches of 4 pairs -> 8 steps, collated by the reference's collator
    assert isinstance(tr.train_dataloader, DevicePrefetcher) and len(tr.train_dataloader) == 8 and tr.eval_dataloader is None
    assert type(tr.train_dataloader.loader.dataset).__module__ == 'align_anything.datasets.text_to_text.preference'
    assert tr.model.total_steps == 8                      # the schedule length comes from the dataloader (supervised_trainer.py:236-239)
    b = next(iter(tr.train_dataloader))
    assert b['input_ids'].shape[0] == 8 and len(b['meta_info']['response_lens']) == 8 and '_window' in b
    assert bool((b['input_ids'][:, -1] == 1).all())       # left padding, every row ends with </s>
    del launches[:]
    hist = tr.train()
    assert len(hist) == 8 and tr.model.global_steps == 8 and 'aa_dpo_loss_fwd_bwd' in launches and hist[-1]['train/epoch'] == 1.0


def test_rm_ppo_grpo_sft_trainers_build_themselves_from_cfgs(launches, tmp_path):
    """The `(cfgs, ds_cfgs)`-only constructors of the other trainers (VERDICT r3 missing #2 / #3 beyond DPO): the score model of RMTrainer from
    model_cfgs.model_name_or_path (text_to_text/rm.py:76-91), the four PPO models from actor / reward / reward_critic paths with the prompt and PTX
    loaders of RLTrainerBase.get_dataloaders (ppo.py:93-154), GRPO's actor / reference / reward (grpo.py:84-139) and SFT's SupervisedDataset
    (sft.py:85-89) -- checkpoints streamed by checkpoint.load_pretrained, datasets by the reference's own plugins on its own asset files."""
    import os
    pref = '/root/reference/assets/text_to_text/preference/train.json'
    sup = '/root/reference/assets/text_to_text/supervised/train.json'
    if not os.path.exists(pref):
        pytest.skip('the reference package (dataset / template plugins) is only present in the build container')
    from oracle import _shim
    _shim.install()
    import transformers as tf
    from tokenizers import Tokenizer, models, pre_tokenizers
    from align_anything_amd.data import DevicePrefetcher
    from align_anything_amd.trainers.grpo import GRPOTrainer
    from align_anything_amd.trainers.ppo import PPOTrainer
    from align_anything_amd.trainers.rm import RMTrainer
    from align_anything_amd.trainers.sft import SupervisedTrainer
    vocab = {w: i for i, w in enumerate(['<s>', '</s>', '<unk>', '<pad>'] + [f'w{i}' for i in range(316)])}
    tk = Tokenizer(models.WordLevel(vocab, unk_token='<unk>'))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    fast = tf.PreTrainedTokenizerFast(tokenizer_object=tk, bos_token='<s>', eos_token='</s>', unk_token='<unk>', pad_token='<pad>')
    fast.chat_template = "{% for m in messages %}{{ m['role'] }} : {{ m['content'] }} </s> {% endfor %}{% if add_generation_prompt %}assistant :{% endif %}"
    torch.manual_seed(0)
    hf = tf.OPTForCausalLM(tf.OPTConfig(hidden_size=128, ffn_dim=256, num_hidden_layers=2, num_attention_heads=2, vocab_size=320, max_position_embeddings=700,
                                        word_embed_proj_dim=128, dropout=0.0, pad_token_id=3)).eval()
    d = str(tmp_path / 'opt')
    hf.save_pretrained(d, max_shard_size='100KB')
    fast.save_pretrained(d)
    data = lambda **kw: dict({'train_size': None, 'train_split': None, 'train_name': None, 'train_data_files': None, 'train_optional_args': [],
                              'eval_datasets': None, 'ptx_datasets': None}, **kw)
    # ---- RM: a language-model checkpoint becomes a score model (fresh head), preference pairs from data_cfgs
    rm = RMTrainer({'train_cfgs': {'per_device_train_batch_size': 8, 'epochs': 1, 'lr_scheduler_type': 'constant'}, 'model_cfgs': {'model_name_or_path': d},
                    'data_cfgs': data(train_datasets=pref, train_template='PKUSafeRLHF')}, {'gradient_clipping': 1.0}, device='cpu')
    assert rm.model.module.kind == 'opt' and rm.tokenizer.padding_side == 'right' and isinstance(rm.train_dataloader, DevicePrefetcher) and len(rm.train_dataloader) == 4
    head = rm.model.module.store.view('score_head.weight')
    assert head.shape == (1, 128) and float(head.float().abs().max()) > 0 and float(head.float().abs().max()) <= 128 ** -0.5 + 1e-3
    assert torch.equal(rm.model.module.state_dict()['model.decoder.layers.0.fc1.weight'].float(), hf.state_dict()['model.decoder.layers.0.fc1.weight'].to(torch.bfloat16).float())
    hist = rm.train()
    assert len(hist) == 4 and rm.model.global_steps == 4 and rm.eval_history == []
    # with data_cfgs.eval_datasets the loop evaluates before the first step, every eval_interval steps and after the epoch (rm.py:274-325)
    rm2 = RMTrainer({'train_cfgs': {'per_device_train_batch_size': 8, 'per_device_eval_batch_size': 4, 'epochs': 1, 'lr_scheduler_type': 'constant', 'eval_strategy': 'steps',
                                    'eval_interval': 2}, 'model_cfgs': {'model_name_or_path': d},
                     'data_cfgs': data(train_datasets=pref, train_template='PKUSafeRLHF', eval_datasets=pref, eval_template='PKUSafeRLHF', eval_size=None, eval_split=None,
                                       eval_name=None, eval_data_files=None, eval_optional_args=[])}, {'gradient_clipping': 1.0}, device='cpu')
    assert len(rm2.eval_dataloader) == 8
    rm2.train()
    assert [s for s, _ in rm2.eval_history] == [0, 2, 4, 4] and set(rm2.eval_history[0][1]) == {'eval/accuracy', 'eval/reward_mean', 'eval/reward_std'}
    # ---- SFT: SupervisedDataset through the same constructor
    sft = SupervisedTrainer({'train_cfgs': {'per_device_train_batch_size': 16, 'epochs': 1, 'lr_scheduler_type': 'constant'}, 'model_cfgs': {'model_name_or_path': d},
                             'data_cfgs': data(train_datasets=sup, train_template='Alpaca')}, {'gradient_clipping': 1.0}, device='cpu')
    assert sft.reference is None and len(sft.train_dataloader) == 2 and type(sft.train_dataloader.loader.dataset).__name__ == 'SupervisedDataset'
    b = next(iter(sft.train_dataloader))
    assert 'labels' in b and b['labels'].shape == b['input_ids'].shape
    # ---- PPO: four models from three paths, prompt + PTX loaders with the RL batch sizes
    ppo_cfgs = {'train_cfgs': {'per_device_prompt_batch_size': 8, 'per_device_train_batch_size': 4, 'epochs': 1, 'update_iters': 1, 'actor_lr_scheduler_type': 'constant',
                               'critic_lr_scheduler_type': 'constant'},
                'model_cfgs': {'actor_model_name_or_path': d, 'reward_model_name_or_path': d, 'reward_critic_model_name_or_path': d, 'model_max_length': 600, 'temperature': 1.0, 'top_p': 1.0},
                'data_cfgs': data(train_datasets=pref, train_template='PKUSafeRLHF', ptx_datasets=sup, ptx_template='Alpaca', ptx_size=None, ptx_split=None, ptx_name=None,
                                  ptx_data_files=None, ptx_optional_args=[])}
    ppo = PPOTrainer(ppo_cfgs, {'gradient_clipping': 1.0}, device='cpu')
    assert ppo.use_ptx and ppo.actor_model.gas == 2 and ppo.reward_model is not None and ppo.reward_critic_model.module.kind == 'opt'
    assert len(ppo.prompt_only_dataloader) == 4 and len(ppo.ptx_dataloader) == 8 and ppo.eval_dataloader is None
    assert type(ppo.prompt_only_dataloader.loader.dataset).__name__ == 'PromptOnlyDataset' and ppo.tokenizer.padding_side == 'left'
    pb = next(iter(ppo.prompt_only_dataloader))
    assert pb['input_ids'].shape[0] == 8 and bool((pb['attention_mask'][:, -1] == 1).all())          # left-padded prompts
    # ---- GRPO: actor / reference / reward, prompts from data_cfgs
    g_cfgs = {'train_cfgs': {'per_device_prompt_batch_size': 16, 'num_generations': 2, 'actor_lr_scheduler_type': 'constant'},
              'model_cfgs': {'actor_model_name_or_path': d, 'reward_model_name_or_path': d}, 'data_cfgs': data(train_datasets=pref, train_template='PKUSafeRLHF')}
    gr = GRPOTrainer(g_cfgs, {'gradient_clipping': 1.0}, device='cpu')
    assert gr.pad_token_id == 3 and gr.eos_token_id == 1 and len(gr.prompt_only_dataloader) == 2 and gr.reward_model is not None
    with pytest.raises(ValueError):
        GRPOTrainer({'train_cfgs': {}, 'model_cfgs': {}}, None, device='cpu')


def test_text_image_trainers_build_themselves_from_cfgs(launches, tmp_path):
    """The headline's OWN modality through the cfgs-only constructors: `DPOTrainer(cfgs, ds_cfgs)` on a LLaVA checkpoint directory (model + a real
    LlavaProcessor, tests/util.tiny_llava_checkpoint) with `data_cfgs.train_template = 'AA_TI2T'` builds the reference's text_image_to_text
    PreferenceDataset / PreferenceCollator (datasets/text_image_to_text/preference.py:77-263) over a local parquet dataset with an Image column:
    batches carry `pixel_values` (the same image for the chosen and the rejected row), the processor expands `<image>` to the tower's 4 image
    tokens, and `train()` runs the epoch through the tower, the projector and the decoder.  `PPOTrainerTI2T(cfgs, ds_cfgs)` builds actor /
    reference / reward / critic from the same directory, prompts through the reference's text_image_to_text PromptOnlyDataset, rolls out with
    the native `generate`, and evaluates on `data_cfgs.eval_datasets` on the reference's schedule (ppo.py:422-485)."""
    import os
    if not os.path.isdir('/root/reference/align_anything'):
        pytest.skip('the reference package (dataset / template plugins) is only present in the build container')
    from oracle import _shim
    _shim.install()
    from tests.util import ti2t_parquet_dataset, tiny_llava_checkpoint
    from align_anything_amd.trainers.dpo import DPOTrainer
    from align_anything_amd.trainers.ppo_ti2t import PPOTrainerTI2T
    d = str(tmp_path / 'llava')
    hf, _ = tiny_llava_checkpoint(d)
    data_dir = ti2t_parquet_dataset(str(tmp_path / 'data'))
    data = lambda **kw: dict({'train_datasets': data_dir, 'train_template': 'AA_TI2T', 'train_size': None, 'train_split': 'train', 'train_name': None,
                              'train_data_files': None, 'train_optional_args': [], 'eval_datasets': None, 'ptx_datasets': None}, **kw)
    cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant', 'weight_decay': 0.0,
                           'per_device_train_batch_size': 4, 'epochs': 1},
            'model_cfgs': {'model_name_or_path': d, 'model_max_length': 256}, 'data_cfgs': data()}
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, device='cpu')
    assert tr.model_cfg['kind'] == 'llava' and type(tr.processor).__name__ == 'LlavaProcessor' and len(tr.train_dataloader) == 3
    assert type(tr.train_dataloader.loader.dataset).__module__ == 'align_anything.datasets.text_image_to_text.preference'
    sd = tr.policy.state_dict()
    hsd = hf.state_dict()
    for tail in ('encoder.layers.1.mlp.fc1.weight', 'multi_modal_projector.linear_2.bias', 'language_model.layers.1.mlp.down_proj.weight', 'lm_head.weight'):
        mine, theirs = [k for k in sd if k.endswith(tail)], [k for k in hsd if k.endswith(tail)]          # key prefixes differ between HF versions
        assert len(mine) == 1 and len(theirs) == 1 and torch.equal(sd[mine[0]].float(), hsd[theirs[0]].to(torch.bfloat16).float()), tail
    b = next(iter(tr.train_dataloader))        # an abandoned iterator: the prefetcher retires its producer before train() starts another
    assert b['input_ids'].shape[0] == 8 and b['pixel_values'].shape == (8, 3, 28, 28) and torch.equal(b['pixel_values'][:4], b['pixel_values'][4:])
    assert bool(((b['input_ids'] == 4).sum(1) == 4).all()) and bool((b['input_ids'][:, -1] == 1).all()) and '_window' in b
    assert len(b['meta_info']['response_lens']) == 8
    del launches[:]
    hist = tr.train()
    assert len(hist) == 3 and tr.model.global_steps == 3
    for k in ('aa_patch_im2col', 'aa_clip_embed', 'aa_image_slot_index', 'aa_dpo_loss_fwd_bwd', 'aa_adamw_flat'):
        assert k in launches, k
    # ---- PPO, text + image: four models from the directory, prompts and evaluation prompts from data_cfgs
    pc = {'train_cfgs': {'per_device_prompt_batch_size': 4, 'per_device_train_batch_size': 4, 'epochs': 1, 'update_iters': 1, 'actor_lr_scheduler_type': 'constant',
                         'critic_lr_scheduler_type': 'constant', 'eval_strategy': 'steps', 'eval_interval': 2},
          'model_cfgs': {'actor_model_name_or_path': d, 'reward_model_name_or_path': d, 'reward_critic_model_name_or_path': d, 'model_max_length': 128,
                         'max_new_tokens': 6, 'temperature': 1.0, 'top_p': 1.0},
          'data_cfgs': data(eval_datasets=data_dir, eval_template='AA_TI2T', eval_size=4, eval_split='train', eval_name=None, eval_data_files=None,
                            eval_optional_args=[])}
    ppo = PPOTrainerTI2T(pc, {'gradient_clipping': 1.0}, device='cpu')
    assert type(ppo.prompt_only_dataloader.loader.dataset).__module__ == 'align_anything.datasets.text_image_to_text.prompt_only'
    assert len(ppo.prompt_only_dataloader) == 3 and len(ppo.eval_dataloader) == 1 and ppo.reward_critic_model.module.kind == 'llava'
    pb = next(iter(ppo.prompt_only_dataloader))
    assert pb['pixel_values'].shape == (4, 3, 28, 28) and bool((pb['attention_mask'][:, -1] == 1).all())
    del launches[:]
    decode = ppo.tokenizer.batch_decode          # the sampler is stubbed out here: "sampled" ids are whatever torch.empty held
    ppo.tokenizer.batch_decode = lambda ids, **kw: decode(ids.clamp(0, 319), **kw)
    # ... and so would be the two kernels whose OUTPUT steers host control flow (the response lengths): stand-ins with defined values
    from align_anything_amd import ops

    def record(selected, unfinished, out, tslot, nact, pad, eos):
        launches.append('aa_decode_record')
        tok = torch.full_like(selected, 7)
        out.scatter_(1, tslot[:, None], tok[:, None])
        return tok

    def pad_left(seq, pad):
        launches.append('aa_move_padding_left')
        rows = [torch.cat([r[r == pad], r[r != pad]]) for r in seq]
        return torch.stack(rows)

    with pytest.MonkeyPatch.context() as mp:
        mp.setattr(ops, 'decode_record', record)
        mp.setattr(ops, 'move_padding_left', pad_left)
        hist = ppo.train()
        assert len(hist) == 3 and 'train/reward_critic_loss' in hist[-1] and ppo.actor_model.global_steps == 3
        # evaluated before the first step and at step 2 (eval_strategy 'steps', interval 2): prompt / completion texts of the 4 evaluation prompts
        assert [s for s, _ in ppo.eval_history] == [0, 2]
        ev = ppo.eval_history[-1][1]
        assert len(ev['eval/prompts']) == 4 and len(ev['eval/generated']) == 4 and all(isinstance(x, str) for x in ev['eval/generated'])
    assert 'aa_attn_decode' in launches and 'aa_ppo_actor_loss' in launches and 'aa_decode_tick' in launches


def test_dpo_step_on_a_llama_with_tied_embeddings(launches):
    """tie_word_embeddings on the Llama family (host flow; the launches are recorded): the head's weight gradient goes into the EMBEDDING's fp32
    gradient buffer with accumulate = 1 (the embedding's own scatter-add lands in the same buffer later in the backward), the optimizer sees one
    parameter, and the frozen reference model carries no second matrix either."""
    from align_anything_amd import configs, ops
    from align_anything_amd.trainers.dpo import DPOTrainer
    z = load_golden('opt_tiny_dpo.npz')
    cfg = configs.llama_cfg(128, 256, 2, 2, 2, 320, rms_eps=1e-5, max_position_embeddings=256, tie_word_embeddings=True)
    seen = []
    real = ops.lmhead_logprob_bwd

    def spy(n, w, labels, lse, dlogp, dw=None, accumulate=False):
        seen.append((w.data_ptr(), None if dw is None else (dw.data_ptr(), dw.dtype, bool(accumulate))))
        return real(n, w, labels, lse, dlogp, dw=dw, accumulate=accumulate)

    tr = DPOTrainer(_cfgs(z), {'gradient_clipping': 1.0}, model_cfg=cfg, device='cpu')
    assert tr.policy.tied and tr.reference.tied and 'lm_head.weight' not in tr.policy.store.specs
    n_untied = DPOTrainer(_cfgs(z), {'gradient_clipping': 1.0}, model_cfg=dict(cfg, tie_word_embeddings=False), device='cpu').policy.parameters_count()
    assert n_untied - tr.policy.parameters_count() == 320 * 128
    with pytest.MonkeyPatch.context() as mp:
        mp.setattr(ops, 'lmhead_logprob_bwd', spy)
        info = tr.train_step(_pref_batch(z))
    assert 'train/loss' in info and tr.model.global_steps == 1 and 'aa_embed_bwd' in launches
    st = tr.policy.store
    assert seen and all(w == st.p[tr.policy.embed].data_ptr() for w, _ in seen)
    assert all(g == (st.g[tr.policy.embed].data_ptr(), torch.float32, True) for _, g in seen)


def test_kto_trainer_builds_its_unmatched_kl_batches_from_cfgs(launches, tmp_path):
    """`KTOTrainer(cfgs, ds_cfgs)` alone: besides the preference loader, `train()` builds the reference's own UnmatchedSupervisedDataset (prompt i
    with the response of sample i - 1, right-padded; kto.py:50-69) over the training set and refreshes the KL estimate from it at the start of
    the epoch (kto.py:211-215) -- `aa_window_kl` once per unmatched batch."""
    import os
    pref = '/root/reference/assets/text_to_text/preference/train.json'
    if not os.path.exists(pref):
        pytest.skip('the reference package (dataset / template plugins) is only present in the build container')
    from oracle import _shim
    _shim.install()
    import transformers as tf
    from tokenizers import Tokenizer, models, pre_tokenizers
    from align_anything_amd.trainers.pref import KTOTrainer
    vocab = {w: i for i, w in enumerate(['<s>', '</s>', '<unk>', '<pad>'] + [f'w{i}' for i in range(316)])}
    tk = Tokenizer(models.WordLevel(vocab, unk_token='<unk>'))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    fast = tf.PreTrainedTokenizerFast(tokenizer_object=tk, bos_token='<s>', eos_token='</s>', unk_token='<unk>', pad_token='<pad>')
    fast.chat_template = "{% for m in messages %}{{ m['role'] }} : {{ m['content'] }} </s> {% endfor %}"
    torch.manual_seed(0)
    hf = tf.OPTForCausalLM(tf.OPTConfig(hidden_size=128, ffn_dim=256, num_hidden_layers=2, num_attention_heads=2, vocab_size=320, max_position_embeddings=700,
                                        word_embed_proj_dim=128, dropout=0.0, pad_token_id=3)).eval()
    d = str(tmp_path / 'opt')
    hf.save_pretrained(d)
    fast.save_pretrained(d)
    cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant', 'per_device_train_batch_size': 8,
                           'per_device_kl_batch_size': 16, 'kl_steps': 1, 'epochs': 1},
            'model_cfgs': {'model_name_or_path': d, 'model_max_length': 600},
            'data_cfgs': {'train_datasets': pref, 'train_template': 'PKUSafeRLHF', 'train_size': None, 'train_split': None, 'train_name': None, 'train_data_files': None,
                          'train_optional_args': [], 'eval_datasets': None}}
    tr = KTOTrainer(cfgs, {'gradient_clipping': 1.0}, device='cpu')
    kl = tr.build_kl_dataloader()
    assert type(kl.loader.dataset).__name__ == 'UnmatchedSupervisedDataset' and len(kl) == 2
    b = next(iter(kl))
    assert b['input_ids'].shape[0] == 16 and len(b['meta_info']['response_lens']) == 16 and '_window' in b
    assert bool((b['input_ids'][:, 0] != 3).all()) and bool((b['input_ids'][:, -1] == 3).any())          # right-padded rows
    del launches[:]
    hist = tr.train()
    assert len(hist) == 4 and launches.count('aa_window_kl') == 2 and 'aa_pref_loss_fwd_bwd' in launches


def test_first_batch_checks_of_the_dpo_trainer(launches, monkeypatch):
    """DPOTrainer._check_first_batch (once, after the first forward): hf's "Image features and image tokens do not match" check is consulted, and a
    batch whose chosen / rejected rows carry DIFFERENT images stops a trainer that runs the tower once per pair (share_vision_tower=True)."""
    from align_anything_amd.trainers.dpo import DPOTrainer
    z = load_golden('llava_tiny_dpo.npz')
    monkeypatch.setenv('AA_VALIDATE_FIRST_BATCH', '1')
    asked = []

    def make(**kw):
        tr = DPOTrainer(_cfgs(z), {'gradient_clipping': 1.0}, model_cfg=tiny_llava_cfg(), policy_state=state_dict_from_golden(z, 'w.', torch.bfloat16),
                        reference_state=state_dict_from_golden(z, 'r.', torch.bfloat16), device='cpu', **kw)
        monkeypatch.setattr(tr.policy, 'validate_batch', lambda: asked.append(1))      # (reads a kernel output that does not exist in a dry run)
        return tr

    tr = make()
    tr.train_step(_pref_batch(z, pixels=True))
    tr.train_step(_pref_batch(z, pixels=True))
    assert asked == [1]                                                               # once per trainer, not per step
    b = _pref_batch(z, pixels=True)
    b['pixel_values'] = b['pixel_values'].clone()
    b['pixel_values'][-1] += 1.0
    with pytest.raises(ValueError, match='share_vision_tower=False'):
        make().train_step(b)
    make(share_vision_tower=False).train_step(b)                                      # every row's image goes through the tower: any layout is fine


def test_ppo_retokenizes_for_a_reward_model_with_its_own_tokenizer(launches, tmp_path):
    """ppo.py:142-143, :226-235: a reward model whose tokenizer differs from the actor's (other vocabulary) is scored on the sequences DECODED with the
    actor's tokenizer and ENCODED with its own (+ eos, padded on its side) -- `batch_retokenize`; with the same tokenizer the ids go through as they are."""
    import transformers as tf
    from tokenizers import Tokenizer, models, pre_tokenizers
    from align_anything_amd.trainers.ppo import PPOTrainer
    words = [f'w{i}' for i in range(316)]

    def save(dirname, vocab_words):
        vocab = {w: i for i, w in enumerate(['<s>', '</s>', '<unk>', '<pad>'] + vocab_words)}
        tk = Tokenizer(models.WordLevel(vocab, unk_token='<unk>'))
        tk.pre_tokenizer = pre_tokenizers.Whitespace()
        fast = tf.PreTrainedTokenizerFast(tokenizer_object=tk, bos_token='<s>', eos_token='</s>', unk_token='<unk>', pad_token='<pad>')
        torch.manual_seed(0)
        hf = tf.OPTForCausalLM(tf.OPTConfig(hidden_size=128, ffn_dim=256, num_hidden_layers=1, num_attention_heads=2, vocab_size=320, max_position_embeddings=128,
                                            word_embed_proj_dim=128, dropout=0.0, pad_token_id=3))
        d = str(tmp_path / dirname)
        hf.save_pretrained(d)
        fast.save_pretrained(d)
        return d

    actor_dir, reward_dir = save('actor', words), save('reward', words[::-1])            # same words, other ids
    cfgs = lambda r: {'train_cfgs': {'actor_lr_scheduler_type': 'constant', 'critic_lr_scheduler_type': 'constant'},
                      'model_cfgs': {'actor_model_name_or_path': actor_dir, 'reward_model_name_or_path': r, 'reward_critic_model_name_or_path': actor_dir, 'model_max_length': 64}}
    same = PPOTrainer(cfgs(actor_dir), {'gradient_clipping': 1.0}, device='cpu')
    assert not same.retokenize_for_reward
    tr = PPOTrainer(cfgs(reward_dir), {'gradient_clipping': 1.0}, device='cpu')
    assert tr.retokenize_for_reward and tr.reward_tokenizer.padding_side == 'right'
    seen = []
    real = tr.reward_model.module.scores
    tr.reward_model.module.scores = lambda ids, am, *a, **k: (seen.append((ids.clone(), am.clone())), real(ids, am, *a, **k))[1]
    ids = torch.tensor([[3, 3, 0, 4, 5, 6, 1], [0, 10, 11, 12, 13, 14, 1]])              # left-padded actor ids: <s> w0 w1 w2 </s>, <s> w6 .. w10 </s>
    am = (ids != 3).long()
    am[0, :2] = 0
    out = tr.reward_model_step(ids, am)
    r_ids, r_am = seen[0]
    assert out['reward'].shape == (2,) and out['reward_values'].shape == (2, 6)
    # decoded without special tokens, re-encoded with the reversed vocabulary + eos, right-padded to the longest
    want0 = [tr.reward_tokenizer.convert_tokens_to_ids(w) for w in ('w0', 'w1', 'w2')] + [1]
    want1 = [tr.reward_tokenizer.convert_tokens_to_ids(w) for w in ('w6', 'w7', 'w8', 'w9', 'w10')] + [1]
    assert r_ids[1].tolist() == want1 and r_ids[0, :4].tolist() == want0 and r_am[0].tolist() == [1, 1, 1, 1, 0, 0] and int(r_ids[0, 0]) == 4 + 315
    with pytest.raises(ValueError, match='critic'):
        PPOTrainer({'train_cfgs': {}, 'model_cfgs': {'actor_model_name_or_path': actor_dir, 'reward_model_name_or_path': actor_dir,
                                                     'reward_critic_model_name_or_path': reward_dir}}, None, device='cpu')


def test_grpo_reward_inputs_equal_the_reference_s_batch_retokenize(launches, tmp_path):
    """grpo.py:229-255: the reward model of the reference NEVER sees the actor's ids -- `compute_rewards` decodes the masked completions (ids after the
    first eos zeroed) without special tokens, appends the reward tokenizer's eos and re-encodes.  The native trainer built from directories does the
    same host work; the ids and mask it hands to the reward model equal the output of the reference's own `batch_retokenize` on the same completions
    (one ends with eos, one was cut at the length cap and gets an eos it never generated, one is empty after masking)."""
    import os
    if not os.path.isdir('/root/reference/align_anything'):
        pytest.skip('the reference package is only present in the build container')
    from oracle import _shim
    _shim.install()
    from align_anything.utils.tools import batch_retokenize
    import transformers as tf
    from tokenizers import Tokenizer, models, pre_tokenizers
    from align_anything_amd.trainers.grpo import GRPOTrainer
    vocab = {w: i for i, w in enumerate(['<pad>', '</s>', '<unk>', '<s>'] + [f'w{i}' for i in range(316)])}
    tk = Tokenizer(models.WordLevel(vocab, unk_token='<unk>'))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    fast = tf.PreTrainedTokenizerFast(tokenizer_object=tk, bos_token='<s>', eos_token='</s>', unk_token='<unk>', pad_token='<pad>')
    torch.manual_seed(0)
    hf = tf.OPTForCausalLM(tf.OPTConfig(hidden_size=128, ffn_dim=256, num_hidden_layers=1, num_attention_heads=2, vocab_size=320, max_position_embeddings=128,
                                        word_embed_proj_dim=128, dropout=0.0, pad_token_id=0, eos_token_id=1, bos_token_id=3))
    d = str(tmp_path / 'm')
    hf.save_pretrained(d)
    fast.save_pretrained(d)
    tr = GRPOTrainer({'train_cfgs': {'num_generations': 1, 'actor_lr_scheduler_type': 'constant'}, 'model_cfgs': {'actor_model_name_or_path': d, 'reward_model_name_or_path': d}},
                     {'gradient_clipping': 1.0}, device='cpu')
    assert tr.reward_tokenizer is not None and tr.pad_token_id == 0 and tr.eos_token_id == 1
    P = 3
    seq = torch.tensor([[3, 10, 11, 20, 21, 1, 0, 0],          # prompt | w16 w17 </s> pad pad
                        [3, 10, 11, 30, 31, 32, 33, 34],       # prompt | five words, no eos: cut at the cap
                        [3, 10, 11, 1, 40, 41, 0, 0]])         # prompt | </s> first: everything after it is masked
    seen = []
    real = tr.reward_model.module.scores
    tr.reward_model.module.scores = lambda ids, am, *a, **k: (seen.append((ids.clone(), am.clone())), real(ids, am, *a, **k))[1]
    from align_anything_amd import ops
    with pytest.MonkeyPatch.context() as mp:      # aa_completion_mask is a kernel: its defined value here = ones up to and including the first eos
        def cmask(c, eos):
            first = torch.where((c == eos).any(1), (c == eos).float().argmax(1), torch.full((c.shape[0],), c.shape[1]))
            return (torch.arange(c.shape[1])[None] <= first[:, None]).to(torch.uint8)
        mp.setattr(ops, 'completion_mask', cmask)
        tr.compute_rewards(seq, P)
    ids, am = seen[0]
    comp = seq[:, P:]
    masked = comp * cmask(comp, 1).to(comp.dtype)
    want = batch_retokenize(masked, src_tokenizer=tr.tokenizer, dest_tokenizer=tr.reward_tokenizer, skip_special_tokens=True, device='cpu')
    assert torch.equal(ids, want['input_ids']) and torch.equal(am, want['attention_mask'])
    assert ids[1].tolist()[-1] == 1 and ids[2].tolist()[0] == 1 and int(am[2].sum()) == 1        # the forced eos; the empty completion is just "</s>"


def test_save_checkpoint_and_load_checkpoint_follow_the_reference(launches, tmp_path):
    """supervised_trainer.py:76-77, :267-268, :435-436: with `train_cfgs.save_checkpoint` a slice also carries the engine's training state; a trainer built
    with `train_cfgs.load_checkpoint` on `model_name_or_path = .../slice_<step>` continues at that step with the saved fp32 masters, Adam moments and
    update count."""
    import os
    ref_assets = '/root/reference/assets/text_to_text/preference/train.json'
    if not os.path.exists(ref_assets):
        pytest.skip('the reference package (dataset / template plugins) is only present in the build container')
    from oracle import _shim
    _shim.install()
    import transformers as tf
    from tokenizers import Tokenizer, models, pre_tokenizers
    from align_anything_amd.trainers.dpo import DPOTrainer
    vocab = {w: i for i, w in enumerate(['<s>', '</s>', '<unk>', '<pad>'] + [f'w{i}' for i in range(316)])}
    tk = Tokenizer(models.WordLevel(vocab, unk_token='<unk>'))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    fast = tf.PreTrainedTokenizerFast(tokenizer_object=tk, bos_token='<s>', eos_token='</s>', unk_token='<unk>', pad_token='<pad>')
    fast.chat_template = "{% for m in messages %}{{ m['role'] }} : {{ m['content'] }} </s> {% endfor %}"
    torch.manual_seed(0)
    hf = tf.OPTForCausalLM(tf.OPTConfig(hidden_size=128, ffn_dim=256, num_hidden_layers=1, num_attention_heads=2, vocab_size=320, max_position_embeddings=600,
                                        word_embed_proj_dim=128, dropout=0.0, pad_token_id=3)).eval()
    d = str(tmp_path / 'opt')
    hf.save_pretrained(d)
    fast.save_pretrained(d)
    out = str(tmp_path / 'run')
    base = lambda path, **t: {'train_cfgs': dict({'scale_coeff': 0.1, 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant', 'per_device_train_batch_size': 8,
                                                  'epochs': 1, 'compute_dtype': 'fp32', 'save_checkpoint': True}, **t),
                              'model_cfgs': {'model_name_or_path': path, 'model_max_length': 512}, 'logger_cfgs': {'output_dir': out, 'save_total_limit': 2},
                              'data_cfgs': {'train_datasets': ref_assets, 'train_template': 'PKUSafeRLHF', 'train_size': None, 'train_split': None, 'train_name': None,
                                            'train_data_files': None, 'train_optional_args': []}}
    tr = DPOTrainer(base(d), {'gradient_clipping': 1.0}, device='cpu')
    st = tr.policy.store
    for g in st.m:                                        # give the moments recognisable contents (the optimizer kernels are not executed here)
        st.m[g].fill_(0.25)
        st.v[g].fill_(0.5)
    tr.train()                                            # 4 steps, a slice every 2
    s2 = os.path.join(out, 'slice_2')
    assert sorted(os.listdir(out)) == ['slice_2', 'slice_4'] and {'pytorch_model.bin', 'native_engine_latest.pt', 'config.json', 'tokenizer.json'} <= set(os.listdir(s2))
    again = DPOTrainer(base(s2, load_checkpoint=True), {'gradient_clipping': 1.0}, device='cpu')
    assert again.global_step == 2 and again.model.global_steps == 2
    for g in again.policy.store.m:
        assert float(again.policy.store.m[g].min()) == 0.25 == float(again.policy.store.m[g].max()) and float(again.policy.store.v[g].min()) == 0.5
    hist = again.train()
    assert len(hist) == 2 and again.global_step == 4      # the epoch's first two batches are skipped (dpo.py:256-270)
    with pytest.raises(ValueError, match='slice_<step>'):
        DPOTrainer(base(d, load_checkpoint=True), {'gradient_clipping': 1.0}, device='cpu')
