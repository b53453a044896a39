"""PPO inner-loop math with the reference's method signatures, executed by the HIP kernels.

Mirrors align_anything/trainers/text_to_text/ppo.py: add_kl_divergence_regularization (:528-547),
get_advantages_and_returns (:487-508), actor_loss_fn (:291-307), critic_loss_fn (:510-526) and
utils/tools.py masked_mean (:460-467).  Each loss returns (loss, grad) -- the fused kernels emit d loss / d input,
which `NativeEngine.backward` consumes the same way the DPO step does (set_pending).  The python loop over t of the
reference's GAE (4*L kernel launches) is one launch here; the per-row `.nonzero()` host syncs of the KL reward are
a device-side last-index scan.

`PPOTrainer` below is the native four-engine `rl_step` (actor train, reference eval, reward eval, critic train) and
`reward_model_step`; only the autoregressive rollout (`generate`) is still the §8(f) "next" row (HIP KV-cache decode).
"""
from __future__ import annotations

import torch

from .. import ops


class PPOMath:
    def __init__(self, kl_coeff=0.02, clip_range_score=50.0, gamma=1.0, gae_lambda=0.95, clip_range_ratio=0.2,
                 clip_range_value=5.0):
        self.kl_coeff, self.clip_range_score = kl_coeff, clip_range_score
        self.gamma, self.gae_lambda = gamma, gae_lambda
        self.clip_range_ratio, self.clip_range_value = clip_range_ratio, clip_range_value

    @staticmethod
    def _m(mask: torch.Tensor) -> torch.Tensor:
        return mask.to(torch.uint8).contiguous()

    def add_kl_divergence_regularization(self, reward, log_probs, ref_log_probs, sequence_mask):
        out, _ = ops.kl_reward(reward.float().contiguous(), log_probs.float().contiguous(),
                               ref_log_probs.float().contiguous(), self._m(sequence_mask), self.kl_coeff,
                               self.clip_range_score)
        return out

    def get_advantages_and_returns(self, values, rewards, sequence_mask, start):
        return ops.gae(values.float().contiguous(), rewards.float().contiguous(), self._m(sequence_mask), int(start),
                       self.gamma, self.gae_lambda)

    def actor_loss_fn(self, log_probs, old_log_probs, advantages, mask, want_grad=True):
        loss, g = ops.ppo_actor_loss(log_probs.float().contiguous(), old_log_probs.float().contiguous(),
                                     advantages.float().contiguous(), self._m(mask), self.clip_range_ratio, want_grad)
        return loss[0], g

    def critic_loss_fn(self, values, old_values, returns, mask, want_grad=True):
        loss, g = ops.ppo_critic_loss(values.float().contiguous(), old_values.float().contiguous(),
                                      returns.float().contiguous(), self._m(mask), self.clip_range_value, want_grad)
        return loss[0], g


def gather_log_probabilities(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """align_anything/utils/tools.py:402-413 drop-in: logits [B, L, V] (bf16/fp32), labels [B, L] -> fp32 [B, L]."""
    B, L, V = logits.shape
    lp, _ = ops.logprob_gather_fwd(logits.reshape(B * L, V), labels.reshape(-1).to(torch.int64).contiguous())
    return lp.view(B, L)


from ..engine import NativeEngine
from ..modeling import build_model
from .common import (build_span_window, cfg_get, eval_due, rl_eval, compute_dtype, end_index, expert_parallel_kwargs, get_all_reduce_max,
                     get_all_reduce_mean, pad_rows, save_interval, save_slice)


class PPOTrainer(PPOMath):
    """Native counterpart of align_anything/trainers/text_to_text/ppo.py::PPOTrainer for the update phase:
    `reward_model_step` (:224-242) and `rl_step` (:309-398) on four DeepSpeedEngine-shaped native engines
    (base/rl_trainer.py:217-272).  The rollout batch (sequences from `generate`) is an input."""

    def __init__(self, cfgs, ds_cfgs=None, *, model_cfg=None, reward_model_cfg=None, actor_state=None, reward_state=None,
                 critic_state=None, device='cuda:0', reward_fn=None, use_ptx=None):
        """reward_fn(input_ids, attention_mask) -> [N] scores replaces the learned reward model: the rule / remote reward of
        trainers/text_to_text/ppo_remote_rm.py:321-347 (decode prompts + responses on the host, score them over HTTP with
        `remote_rm_client.score`; that string work stays the caller's Python) -- the critic still comes from `reward_model_cfg`.
        The phases are the reference's own methods in its order (text_to_text/ppo.py:62-91): init_check, init_models, init_datasets,
        init_engines, init_logger -- a modality subclass overrides `init_models` / `init_datasets` as the reference's do."""
        self.cfgs, self.ds_train_cfgs, self.device = cfgs, ds_cfgs, torch.device(device)
        self.model_cfg, self.reward_model_cfg, self.reward_fn = model_cfg, reward_model_cfg, reward_fn
        self._use_ptx_arg = use_ptx
        self.tokenizer = self.processor = self.hf_config = None
        self.prompt_only_dataloader = self.eval_dataloader = self.ptx_dataloader = None
        self.global_step = 0
        self.init_check()
        self.init_models(actor_state, reward_state, critic_state)
        self.init_datasets()
        self.init_engines()
        self.init_logger()

    # ------------------------------------------------------------------ init_* (ppo.py:62-207, base/rl_trainer.py:217-272)
    def init_check(self) -> None:
        from .common import refuse_unsupported_options
        refuse_unsupported_options(self.cfgs)
        t = lambda k, d: cfg_get(self.cfgs, 'train_cfgs.' + k, d)
        PPOMath.__init__(self, kl_coeff=float(t('kl_coeff', 0.02)), clip_range_score=float(t('clip_range_score', 50.0)),
                         gamma=float(t('gamma', 1.0)), gae_lambda=float(t('gae_lambda', 0.95)),
                         clip_range_ratio=float(t('clip_range_ratio', 0.2)), clip_range_value=float(t('clip_range_value', 5.0)))
        self.ptx_coeff = float(t('ptx_coeff', 16.0))        # configs/train/text_to_text/ppo.yaml:71
        pb, tb = t('per_device_prompt_batch_size', None), t('per_device_train_batch_size', None)
        if pb is not None and tb is not None and int(pb) % int(tb) != 0:      # ppo.py:172-182
            raise ValueError('The number of prompt-only samples must be divisible by the micro batch size.')
        self._from_paths = self.model_cfg is None
        if self._from_paths:
            m_ = lambda k: cfg_get(self.cfgs, 'model_cfgs.' + k, None)
            if not m_('actor_model_name_or_path'):
                raise ValueError('PPOTrainer: model_cfg or model_cfgs.actor_model_name_or_path is required')
            self._paths = {'actor': m_('actor_model_name_or_path'), 'reward': m_('reward_model_name_or_path'),
                           'critic': m_('reward_critic_model_name_or_path') or m_('reward_model_name_or_path')}
            if self.reward_fn is None and not self._paths['reward']:
                raise ValueError('PPOTrainer: model_cfgs.reward_model_name_or_path (or a reward_fn) is required')

    def init_models(self, actor_state=None, reward_state=None, critic_state=None) -> None:
        """ppo.py:93-147: actor (trainable, left padding), reference (frozen, same checkpoint), reward model (frozen score model, right padding;
        absent with a reward_fn), reward critic (trainable score model, from reward_critic_model_name_or_path or the reward model's directory)."""
        cfgs, device = self.cfgs, self.device
        dt = compute_dtype(cfg_get(cfgs, 'train_cfgs.compute_dtype', 'bf16'))   # fp32 = parity mode for the update phase (rollouts need bf16)
        model_cfg, reward_fn = self.model_cfg, self.reward_fn
        if self._from_paths:
            # `PPOTrainer(cfgs, ds_cfgs)` alone, as the reference's constructor: geometry of the four models from the config.json under
            # model_cfgs.{actor,reward,reward_critic}_model_name_or_path; the weights are streamed in below
            from transformers import AutoConfig
            from .. import configs as _configs
            model_cfg = _configs.from_hf_config(AutoConfig.from_pretrained(self._paths['actor'], trust_remote_code=True))
            rp = self._paths['critic'] or self._paths['actor']
            self.reward_model_cfg = _configs.from_hf_config(AutoConfig.from_pretrained(rp, trust_remote_code=True))
        rcfg = self.reward_model_cfg or model_cfg
        epk = expert_parallel_kwargs(cfgs, model_cfg)            # train_cfgs.expert_parallel on a Qwen3-MoE actor (see trainers/grpo.py)
        rpk = epk if rcfg.get('kind') == 'qwen3moe' else {}
        if self._from_paths:
            from ..checkpoint import load_pretrained
            mml = int(cfg_get(cfgs, 'model_cfgs.model_max_length', 512))
            actor, self.tokenizer, self.processor, self.hf_config = load_pretrained(self._paths['actor'], device, trainable=True, dtype=dt, model_max_length=mml,
                                                                                    padding_side='left', build_kwargs=epk)
            ref = load_pretrained(self._paths['actor'], device, trainable=False, dtype=dt, model_max_length=mml, padding_side='left', build_kwargs=epk)[0]
            reward = None
            if reward_fn is None:
                reward, self.reward_tokenizer, _, _ = load_pretrained(self._paths['reward'], device, trainable=False, head='score', dtype=dt, model_max_length=mml,
                                                                      padding_side='right', build_kwargs=rpk)
            critic, self.reward_critic_tokenizer, _, _ = load_pretrained(self._paths['critic'] or self._paths['actor'], device, trainable=True, head='score', dtype=dt,
                                                                         model_max_length=mml, padding_side='left', build_kwargs=rpk)
            model_cfg, rcfg = actor.cfg, critic.cfg
            # ppo.py:142-143 `is_same_tokenizer` (same class and vocabulary): otherwise reward_model_step re-tokenises, as the reference does.  The
            # critic reads the ACTOR's ids (ppo.py:237-239), so its tokenizer must be the actor's.
            same = lambda a, b: a is b or (a.__class__ == b.__class__ and a.get_vocab() == b.get_vocab())
            rt = getattr(self, 'reward_tokenizer', None)
            self.retokenize_for_reward = bool(reward is not None and rt is not None and self.tokenizer is not None and not same(self.tokenizer, rt))
            ct = getattr(self, 'reward_critic_tokenizer', None)
            if ct is not None and self.tokenizer is not None and not same(self.tokenizer, ct):
                raise ValueError('PPOTrainer: the reward critic scores the actor\'s token ids (ppo.py:237-239); its checkpoint must share the actor\'s tokenizer')
        else:
            actor = build_model(model_cfg, device, trainable=True, dtype=dt, **epk)
            ref = build_model(model_cfg, device, trainable=False, dtype=dt, **epk)
            reward = build_model(rcfg, device, trainable=False, head='score', dtype=dt, **rpk) if reward_fn is None else None
            critic = build_model(rcfg, device, trainable=True, head='score', dtype=dt, **rpk)
        if actor_state is not None:
            actor.load_state_dict(actor_state)
            ref.load_state_dict(actor_state)
        if reward_state is not None and reward is not None:
            reward.load_state_dict(reward_state)
        if critic_state is not None or reward_state is not None:
            critic.load_state_dict(critic_state if critic_state is not None else reward_state)
        self.model_cfg, self.reward_model_cfg = model_cfg, rcfg
        self._modules = {'actor': actor, 'ref': ref, 'reward': reward, 'critic': critic}

    def init_engines(self) -> None:
        cfgs, ds_cfgs = self.cfgs, self.ds_train_cfgs
        t = lambda k, d: cfg_get(cfgs, 'train_cfgs.' + k, d)
        mods = self._modules
        clip = float(cfg_get(ds_cfgs, 'gradient_clipping', 1.0))
        betas = [float(b) for b in t('actor_betas', t('adam_betas', [0.9, 0.95]))]
        # base/rl_trainer.py:217-260: the schedule length comes from the prompt dataloader, which train() may still receive -> the engines start
        # with an unknown total (explicit train_cfgs.total_training_steps = number of MICRO steps wins) and train() fills it in.
        # With a PTX dataset the actor accumulates over (rl_step, ptx_step) pairs: its accumulation depth and micro-step total double
        # (:231-234), so both losses are scaled by 1 / (2 gas) and land in ONE optimizer update.
        self.use_ptx = bool(cfg_get(cfgs, 'data_cfgs.ptx_datasets', None)) if self._use_ptx_arg is None else bool(self._use_ptx_arg)
        self.gas = int(cfg_get(ds_cfgs, 'gradient_accumulation_steps', t('gradient_accumulation_steps', 1)))
        total = t('total_training_steps', None)
        a_gas = self.gas * (2 if self.use_ptx else 1)
        a_total = None if total is None else max(1, int(total) * (2 if self.use_ptx else 1) // a_gas)
        c_total = None if total is None else max(1, int(total) // self.gas)
        self.actor_model = NativeEngine(mods['actor'], lr=float(t('actor_lr', 1e-5)), betas=betas, weight_decay=float(t('actor_weight_decay', 0.01)),
                                        max_grad_norm=clip, total_steps=a_total, warmup_steps=int(float(t('actor_lr_warmup_ratio', 0.03)) * (a_total or 0)),
                                        lr_scheduler_type=t('actor_lr_scheduler_type', 'cosine'), gradient_accumulation_steps=a_gas)
        self.reward_critic_model = NativeEngine(mods['critic'], lr=float(t('critic_lr', 5e-6)), betas=betas, weight_decay=float(t('critic_weight_decay', 0.0)),
                                                max_grad_norm=clip, total_steps=c_total, warmup_steps=int(float(t('critic_lr_warmup_ratio', 0.03)) * (c_total or 0)),
                                                lr_scheduler_type=t('critic_lr_scheduler_type', 'constant'), gradient_accumulation_steps=self.gas)
        self.actor_reference_model = NativeEngine(mods['ref'], trainable=False)
        self.reward_model = NativeEngine(mods['reward'], trainable=False) if mods['reward'] is not None else None
        self._modules = None

    def init_logger(self) -> None:
        self.logger = None          # observability is out of scope; train() returns the metrics

    def set_train(self, mode: bool = True) -> None:
        """ppo.py:400-408 / base/rl_trainer.py:274-286: training mode for the two trainable engines (the native models carry no dropout; the
        flag is kept for callers that branch on it)."""
        self.actor_model.module.train(mode)
        self.reward_critic_model.module.train(mode)

    @staticmethod
    def split_ptx_micro_batches(ptx_batch) -> list:
        """ppo.py:195-207: a PTX batch becomes ONE-row micro-batches (`split_ptx_micro_batches` slices with micro_batch_size = 1 ... in
        steps of per_device_train_batch_size; the reference's quirk kept: one row each)."""
        return [{k: (v[i:i + 1] if isinstance(v, torch.Tensor) else v) for k, v in ptx_batch.items()} for i in range(ptx_batch['input_ids'].shape[0])]

    def init_datasets(self) -> None:
        """ppo.py:149-154 `get_dataloaders(PromptOnlyDataset, PromptOnlyDataset, SupervisedDataset)` through the reference's own dataset / template
        plugins (common.get_dataloaders, RL batch sizes); `train()` falls back to these loaders when called without arguments."""
        if not self._from_paths:
            return               # injected models carry no tokenizer: the caller hands train() its dataloaders
        from .common import get_dataloaders
        self.pad_token_id = cfg_get(self.cfgs, 'model_cfgs.pad_token_id', getattr(self.tokenizer, 'pad_token_id', None))
        self.prompt_only_dataloader, self.eval_dataloader, self.ptx_dataloader = get_dataloaders(self, 'PromptOnlyDataset', 'PromptOnlyDataset',
                                                                                                 ptx_dtype_name='SupervisedDataset', rl=True)

    def _token_id(self, key, default):
        """pad / eos id of the rollouts: model_cfgs.<key> when configured, else the tokenizer's (the cfgs-only constructor loaded it: the reference
        takes both from the tokenizer, ppo.py:161-170), else `default`."""
        v = cfg_get(self.cfgs, 'model_cfgs.' + key, None)
        if v is None:
            v = getattr(getattr(self, 'tokenizer', None), key, None)
        return default if v is None else int(v)

    # ------------------------------------------------------------------ rollout (ppo.py:209-222, 244-289)
    @ops.few_row_gemms
    def actor_step(self, prompt_batch, generator=None, sequences=None):
        """`self.actor_model.module.generate(**batch, generation_config=..., do_sample=True)` natively (ppo.py:209-222).
        `sequences` injects already generated rows (tests against the reference's fixture, external samplers)."""
        from ..generation import generate
        m = lambda k, d: cfg_get(self.cfgs, 'model_cfgs.' + k, d)
        pad = self._token_id('pad_token_id', 0)
        if sequences is not None:
            return {'input_ids': sequences, 'attention_mask': sequences.ne(pad)}
        self.actor_model.wait_optimizer()
        seq = generate(self.actor_model.module, prompt_batch['input_ids'], prompt_batch['attention_mask'],
                       max_length=int(m('model_max_length', 2048)), do_sample=True, temperature=float(m('temperature', 1.0)),
                       top_p=float(m('top_p', 1.0)), top_k=m('top_k', 'hf'), repetition_penalty=float(m('repetition_penalty', 1.0)),
                       eos_token_id=self._token_id('eos_token_id', None), pad_token_id=pad,
                       pixel_values=prompt_batch.get('pixel_values'), generator=generator)
        return {'input_ids': seq, 'attention_mask': seq.ne(pad)}

    @ops.few_row_gemms
    def rollout(self, prompt_only_batch, generator=None, sequences=None):
        """One micro-batch of experience: generate, score, log-probs of actor and reference (all no-grad)."""
        actor_batch = self.actor_step(prompt_only_batch, generator, sequences)
        ids, am = actor_batch['input_ids'], actor_batch['attention_mask'].to(torch.int64)
        scored = self.reward_model_step(ids, am)
        log_probs, _ = self.sequence_log_probs(self.actor_model, ids, am, 0)
        ref_log_probs, _ = self.sequence_log_probs(self.actor_reference_model, ids, am, 0)
        training = {'prompt_idx': prompt_only_batch['input_ids'].size(-1) - 1, 'log_probs': log_probs,
                    'ref_log_probs': ref_log_probs, 'reward': scored['reward'], 'reward_values': scored['reward_values']}
        return {'input_ids': ids, 'attention_mask': am}, training

    # ------------------------------------------------------------------ scoring (ppo.py:224-242)
    def reward_model_step(self, input_ids, attention_mask):
        """reward = end score of the reward model (last attended token, models/opt.py:67-89);
        reward_values = critic scores[:, :-1]."""
        N, T = input_ids.shape
        if self.reward_fn is not None:     # ppo_remote_rm.py:328-340: whatever the scorer returns, as a 1-D tensor on the device
            reward = torch.as_tensor(self.reward_fn(input_ids, attention_mask)).detach().to(device=input_ids.device, dtype=torch.float32)
            reward = reward.reshape(1) if reward.dim() == 0 else reward
            if reward.shape != (N,):
                raise ValueError(f'reward_fn returned shape {tuple(reward.shape)}, expected ({N},)')
        else:
            r_ids, r_am = input_ids, attention_mask
            if getattr(self, 'retokenize_for_reward', False):
                # ppo.py:226-235 `batch_retokenize`: the reward model has a tokenizer of its own -> the sequences are decoded with the actor's and
                # encoded with the reward model's (+ its eos, padded to the longest on the reward tokenizer's side); host string work, as in the reference.
                # (The reference then ALSO hands the re-tokenised ids to rl_step as the actor's input, ppo.py:279 -- with the actor's mask, whose shape no
                # longer fits: a defect of that rarely used path that is not reproduced; the update runs on the actor's own ids.)
                texts = self.tokenizer.batch_decode(input_ids, skip_special_tokens=True)
                enc = self.reward_tokenizer([t + self.reward_tokenizer.eos_token for t in texts], padding=True, truncation=False, return_tensors='pt')
                r_ids, r_am = enc['input_ids'].to(input_ids.device), enc['attention_mask'].to(input_ids.device)
            scores = self.reward_model.module.scores(r_ids, r_am)
            end = end_index(self.reward_model.module.kind, r_am)
            reward = scores[torch.arange(N, device=scores.device), end]
        self.reward_critic_model.wait_optimizer()
        values = self.reward_critic_model.module.scores(input_ids, attention_mask)[:, :-1]
        return {'reward': reward, 'reward_values': values}

    def sequence_log_probs(self, engine, input_ids, attention_mask, start=0, save=False):
        """gather_log_probabilities(logits[:, :-1], input_ids[:, 1:])[:, start:] -> fp32 [B, T-1-start]."""
        w = build_span_window(input_ids, start)
        if hasattr(engine, 'wait_optimizer'):
            engine.wait_optimizer()
        lp = engine.module.response_logprobs(input_ids, attention_mask, w, save=save)
        return lp[:w['rows']].view(w['N'], w['W']), w

    # ------------------------------------------------------------------ PTX mix-in (ppo.py:400-408)
    @ops.few_row_gemms
    def ptx_step(self, ptx_batch):
        """One supervised micro-step on the actor with the pre-training / SFT batch (`input_ids`, `labels`, `attention_mask`):
        backward of ptx_coeff * (HF causal-LM loss), logged unscaled -- the native form of the supervised loss is trainers/sft.py.
        With `use_ptx` the actor engine accumulates over 2 x gas micro-steps (base/rl_trainer.py:231-234), so the preceding
        rl_step's `step()` was a no-op and the `step()` here applies the sum of both gradients as one update."""
        from .common import build_label_window
        ids = ptx_batch['input_ids']
        w = ptx_batch.get('_window') or build_label_window(ptx_batch['labels'], device=ids.device)
        self.actor_model.wait_optimizer()
        logp = self.actor_model.module.response_logprobs(ids, ptx_batch.get('attention_mask'), w, save=True)
        ptx_loss, dlogp = ops.sft_loss(logp, w['rows'])
        self.actor_model.set_pending(dlogp * self.ptx_coeff)
        self.actor_model.backward(ptx_loss)
        self.actor_model.step()
        return {'train/ptx_loss': float(get_all_reduce_mean(ptx_loss.reshape(1).clone()).item())}

    # ------------------------------------------------------------------ outer loop (ppo.py:410-480)
    def _rollout_micro_batch(self, per_device_train_batch_size: int) -> int:
        """Rows per rollout micro-batch (ppo.py:244-289 splits the prompt batch); 0 = roll out the whole prompt batch at once."""
        return per_device_train_batch_size

    def _set_schedules(self, prompt_only_dataloader, use_ptx: bool) -> None:
        """base/rl_trainer.py:217-260: total micro-steps = len(prompt dataloader) x epochs x update_iters x per_device_train_batch_size
        x per_device_prompt_batch_size; the schedulers run over total // gas updates; with PTX the actor's gas and total double."""
        t = lambda k, d: cfg_get(self.cfgs, 'train_cfgs.' + k, d)
        a, c = self.actor_model, self.reward_critic_model
        if use_ptx != self.use_ptx:
            if a.global_steps:
                raise RuntimeError('PTX dataloader presence changed after the first actor update')
            self.use_ptx = use_ptx
        a_gas = self.gas * (2 if use_ptx else 1)
        explicit = t('total_training_steps', None)
        if explicit is None and not hasattr(prompt_only_dataloader, '__len__'):
            if a.gas != a_gas and not a.global_steps and a.micro_steps % a.gas == 0:
                a.gas, a.micro_steps = a_gas, 0
            return           # unsized iterable and no explicit total: a cosine engine will refuse to step (engine._lr_at)
        total = int(explicit) if explicit is not None else (
            len(prompt_only_dataloader) * int(t('epochs', 1)) * int(t('update_iters', 1)) * int(t('per_device_train_batch_size', 8))
            * int(t('per_device_prompt_batch_size', 1)))
        # A second train() call keeps the schedule it started with; an engine RESUMED through load_checkpoint under a decaying schedule has steps behind it
        # but no schedule length yet (total_steps None) and must still learn it, or its next step() raises (the RM / DPO / GRPO fix, here too)
        if not (a.global_steps and a.total_steps is not None):
            a.set_schedule(max(1, total * (2 if use_ptx else 1) // a_gas), float(t('actor_lr_warmup_ratio', 0.03)), a_gas)
        if not (c.global_steps and c.total_steps is not None):
            c.set_schedule(max(1, total // self.gas), float(t('critic_lr_warmup_ratio', 0.03)), self.gas)

    @staticmethod
    def _rows(batch, lo, hi):
        return {k: (v[lo:hi] if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}

    def train(self, prompt_only_dataloader=None, ptx_dataloader=None, generator=None):
        """The reference's PPO loop without its logging / checkpoint plumbing: every prompt batch is rolled out in micro-batches of
        `per_device_train_batch_size` prompts (ppo.py:244-289), then `update_iters` passes of `rl_step` (+ `ptx_step` when a PTX
        dataloader is given) run over those micro-batches.  Reference quirks kept: PTX batches are cycled to the length of the prompt set
        (:427-433), split into ONE-row micro-batches (`split_ptx_micro_batches`, :195-207) and zipped with the rollout micro-batches (the
        shorter list wins).  Returns the per-step metric dicts."""
        import itertools
        t = lambda k, d: cfg_get(self.cfgs, 'train_cfgs.' + k, d)
        if prompt_only_dataloader is None:      # the loaders init_datasets() built from data_cfgs (the cfgs-only constructor)
            prompt_only_dataloader = getattr(self, 'prompt_only_dataloader', None)
            ptx_dataloader = ptx_dataloader if ptx_dataloader is not None else getattr(self, 'ptx_dataloader', None)
            if prompt_only_dataloader is None:
                raise ValueError('PPOTrainer.train needs a prompt dataloader (argument, or data_cfgs.train_datasets with the cfgs-only constructor)')
        epochs, update_iters = int(t('epochs', 1)), int(t('update_iters', 1))
        micro = self._rollout_micro_batch(int(t('per_device_train_batch_size', 8)))
        use_ptx = ptx_dataloader is not None
        self._set_schedules(prompt_only_dataloader, use_ptx)
        self.global_step = getattr(self, 'global_step', 0)
        history = []
        # ppo.py:462-468: slice_<global_step> every total_update_steps // save_total_limit steps (rl_trainer.py:217-227: total_update_steps =
        # len(prompt dataloader) x epochs x update_iters x per_device_train_batch_size x per_device_prompt_batch_size)
        every = save_interval(self.cfgs, len(prompt_only_dataloader) * epochs * update_iters * int(t('per_device_train_batch_size', 8))
                              * int(t('per_device_prompt_batch_size', 1)) if hasattr(prompt_only_dataloader, '__len__') else None)
        self.eval_history = getattr(self, 'eval_history', [])
        if eval_due(self.cfgs, 'begin'):                                                  # ppo.py:422-424
            self.eval_history.append((self.global_step, self.eval()))
        for _ in range(epochs):
            ptx_iter = itertools.cycle(ptx_dataloader) if use_ptx else None
            for prompt_batch in prompt_only_dataloader:
                n = prompt_batch['input_ids'].shape[0]
                step = micro or n
                rollouts = [self.rollout(self._rows(prompt_batch, i, i + step), generator) for i in range(0, n, step)]
                if use_ptx:
                    pb = next(ptx_iter)
                    ptx_batches = self.split_ptx_micro_batches(pb)
                else:
                    ptx_batches = [None] * len(rollouts)
                for _ in range(update_iters):
                    for (inference_batch, training_batch), ptx_batch in zip(rollouts, ptx_batches):
                        info = self.rl_step(inference_batch, training_batch)
                        if use_ptx:
                            info.update(self.ptx_step(ptx_batch))
                        self.global_step += 1
                        history.append(info)
                        if every and self.global_step % every == 0:
                            self.save(tag=self.global_step)
                        if eval_due(self.cfgs, 'steps', self.global_step):                # ppo.py:471-479
                            self.eval_history.append((self.global_step, self.eval()))
            if eval_due(self.cfgs, 'epoch'):                                              # ppo.py:481-485
                self.eval_history.append((self.global_step, self.eval()))
        return history

    def eval(self, eval_dataloader=None) -> dict:
        """base/rl_trainer.py:289-329 (common.rl_eval): actor completions for the evaluation prompts."""
        return rl_eval(self, eval_dataloader)

    def save(self, model=None, tag=None, output_dir=None) -> str:
        """ppo.py:549-555 / base/rl_trainer.py save_transformers: the ACTOR in the layout `from_pretrained` loads (common.save_slice)."""
        return save_slice(self, model or self.actor_model, tag, output_dir)

    # ------------------------------------------------------------------ update (ppo.py:309-398)
    @ops.few_row_gemms
    def rl_step(self, inference_batch, training_batch):
        old_log_probs = training_batch['log_probs'].float()
        ref_log_probs = training_batch['ref_log_probs'].float()
        reward = training_batch['reward'].float()
        old_reward_values = training_batch['reward_values'].float()
        start = int(training_batch['prompt_idx'])
        input_ids, attention_mask = inference_batch['input_ids'], inference_batch['attention_mask']
        sequence_mask = attention_mask[:, 1:].bool()
        mask_s = sequence_mask[:, start:].contiguous()

        old_rewards = self.add_kl_divergence_regularization(reward, old_log_probs, ref_log_probs, sequence_mask)
        reward_advantages, reward_returns = self.get_advantages_and_returns(old_reward_values, old_rewards, sequence_mask, start)

        log_probs, w = self.sequence_log_probs(self.actor_model, input_ids, attention_mask, start, save=True)
        actor_loss, dlogp = self.actor_loss_fn(log_probs.contiguous(), old_log_probs[:, start:].contiguous(),
                                               reward_advantages, mask_s)
        self.actor_model.set_pending(pad_rows(dlogp, w['rows_pad']))
        self.actor_model.backward(actor_loss)
        self.actor_model.step()

        self.reward_critic_model.wait_optimizer()
        values = self.reward_critic_model.module.response_scores(input_ids, attention_mask, w, save=True)
        reward_values = values[:w['rows']].view(w['N'], w['W'])
        critic_loss, dvalues = self.critic_loss_fn(reward_values.contiguous(), old_reward_values[:, start:].contiguous(),
                                                   reward_returns, mask_s)
        self.reward_critic_model.set_pending(pad_rows(dvalues, w['rows_pad']))
        self.reward_critic_model.backward(critic_loss)
        self.reward_critic_model.step()

        m = mask_s.float()
        cnt = m.sum(-1)
        mm = lambda x: ((x * m).sum(-1) / cnt).mean()
        stats = torch.stack([actor_loss, critic_loss, reward.mean(), (old_rewards[:, start:] * m).sum(-1).mean(),
                             mm(reward_advantages), mm(reward_returns), mm(reward_values),
                             ((old_log_probs - ref_log_probs)[:, start:] * m).sum(-1).mean(), cnt.mean()])
        stats = get_all_reduce_mean(stats)
        mx = get_all_reduce_max(cnt.max().reshape(1))
        s = stats.tolist()
        return {
            'train/actor_loss': s[0], 'train/reward_critic_loss': s[1], 'train/reward': s[2],
            'train/reward_with_kl_penalty': s[3], 'train/reward_advantage': s[4], 'train/reward_return': s[5],
            'train/reward_value': s[6], 'train/kl_divergence': s[7],
            'train/actor_lr': self.actor_model.optimizer.param_groups[0]['lr'],
            'train/reward_critic_lr': self.reward_critic_model.optimizer.param_groups[0]['lr'],
            'train/mean_generated_length': s[8], 'train/max_generated_length': float(mx.item()),
        }
