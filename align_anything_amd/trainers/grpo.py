"""Native GRPO step: align_anything/trainers/text_to_text/grpo.py::GRPOTrainer (train_step :257-329) on the MI355X engines.

Per step: G sampled completions per prompt (`generate_completions` :212-227), a reward per completion from a score
model on the completion truncated after its first EOS (`compute_rewards` :229-255), group-normalised advantages,
per-token log-probs of actor and reference on the completion window (`_get_per_token_logps` :199-210), the k3-KL
regularised loss (:291-316), backward and optimizer step.  All tensor math runs in the HIP kernels of rl_math.hip
(`aa_group_advantage`, `aa_completion_mask`, `aa_grpo_loss_fwd_bwd`); there is no torch fallback.
"""
from __future__ import annotations

import torch

from .. import ops
from ..engine import NativeEngine
from ..modeling import build_model
from .common import build_span_window, cfg_get, eval_due, rl_eval, compute_dtype, end_index, get_all_reduce_mean, pad_rows, expert_parallel_kwargs, save_interval, save_slice


class GRPOTrainer:
    """Three engines as in grpo.py:153-196: trainable actor, frozen actor reference, frozen reward (score) model.
    The reference re-tokenises completions for the reward model (`batch_retokenize`, grpo.py:243-250): so does `compute_rewards` when the trainer
    loaded its tokenizers (the cfgs-only constructor); built from injected models it scores the masked completion ids directly (`reward_fn` overrides both)."""

    def __init__(self, cfgs, ds_cfgs=None, *, model_cfg=None, reward_model_cfg=None, actor_state=None, reference_state=None,
                 reward_state=None, reward_fn=None, device='cuda:0'):
        """`GRPOTrainer(cfgs, ds_cfgs)` alone, as the reference's constructor (text_to_text/grpo.py:55-75): actor / reference from
        model_cfgs.actor_model_name_or_path, reward model from reward_model_name_or_path (grpo.py:84-133), prompts from data_cfgs (:135-139);
        the keyword arguments inject pre-built pieces.  Phases = the reference's methods in its order (grpo.py:69-75)."""
        self.cfgs, self.ds_train_cfgs, self.device = cfgs, ds_cfgs, torch.device(device)
        self.model_cfg, self.reward_model_cfg, self.reward_fn = model_cfg, reward_model_cfg, reward_fn
        self.tokenizer = self.reward_tokenizer = self.processor = self.hf_config = None
        self.prompt_only_dataloader = self.eval_dataloader = None
        self.global_step = 0
        self.init_check()
        self.init_models(actor_state, reference_state, reward_state)
        self.init_datasets()
        self.init_engines()
        self.init_logger()

    # ------------------------------------------------------------------ init_* (grpo.py:69-196)
    def init_check(self) -> None:
        from .common import refuse_unsupported_options
        refuse_unsupported_options(self.cfgs)
        cfgs = self.cfgs
        t = lambda k, d: cfg_get(cfgs, 'train_cfgs.' + k, d)
        self._from_paths = self.model_cfg is None
        if self._from_paths:
            self._paths = (cfg_get(cfgs, 'model_cfgs.actor_model_name_or_path', None), cfg_get(cfgs, 'model_cfgs.reward_model_name_or_path', None))
            if not self._paths[0] or (self.reward_fn is None and not self._paths[1]):
                raise ValueError('GRPOTrainer: model_cfg, or model_cfgs.actor_model_name_or_path (+ reward_model_name_or_path or a reward_fn), is required')
        self.beta = float(t('beta', 0.04))                       # grpo.py:67 default
        self.num_generations = int(t('num_generations', 4))      # grpo.py:66
        self.pad_token_id = int(cfg_get(cfgs, 'model_cfgs.pad_token_id', 0))
        self.eos_token_id = int(cfg_get(cfgs, 'model_cfgs.eos_token_id', 2))

    def init_models(self, actor_state=None, reference_state=None, reward_state=None) -> None:
        cfgs, device = self.cfgs, self.device
        model_cfg, reward_model_cfg = self.model_cfg, self.reward_model_cfg
        dt = compute_dtype(cfg_get(cfgs, 'train_cfgs.compute_dtype', 'bf16'))   # fp32 = parity mode (sequences / rewards must then be injected)
        if self._from_paths:
            from transformers import AutoConfig
            from .. import configs as _configs
            ap_, rp_ = self._paths
            model_cfg = _configs.from_hf_config(AutoConfig.from_pretrained(ap_, trust_remote_code=True))
            reward_model_cfg = _configs.from_hf_config(AutoConfig.from_pretrained(rp_, trust_remote_code=True)) if rp_ else None
        # train_cfgs.expert_parallel (Qwen3-MoE, BASELINE configs[4]): actor, reference and reward model hold 1/world of the experts;
        # the rollout then exchanges tokens per decode position with every rank stepping in lockstep (generation.py)
        epk = expert_parallel_kwargs(cfgs, model_cfg)
        rpk = epk if (reward_model_cfg or model_cfg).get('kind') == 'qwen3moe' else {}
        reward = None
        if self._from_paths:
            from ..checkpoint import load_pretrained
            mml = int(cfg_get(cfgs, 'model_cfgs.model_max_length', 512))
            actor, self.tokenizer, self.processor, self.hf_config = load_pretrained(ap_, device, trainable=True, dtype=dt, model_max_length=mml, padding_side='left',
                                                                                    build_kwargs=epk)
            ref = load_pretrained(ap_, device, trainable=False, dtype=dt, model_max_length=mml, padding_side='left', build_kwargs=epk)[0]
            if self.tokenizer is not None:
                # pad and eos are resolved INDEPENDENTLY (as PPOTrainer._token_id): the yaml's value when set, else the tokenizer's
                # (grpo.py takes both from the tokenizer), else the default -- a yaml that sets only pad_token_id must not freeze eos at 2
                if cfg_get(cfgs, 'model_cfgs.pad_token_id', None) is None and self.tokenizer.pad_token_id is not None:
                    self.pad_token_id = int(self.tokenizer.pad_token_id)
                if cfg_get(cfgs, 'model_cfgs.eos_token_id', None) is None and self.tokenizer.eos_token_id is not None:
                    self.eos_token_id = int(self.tokenizer.eos_token_id)
            if self.reward_fn is None:
                reward, self.reward_tokenizer, _, _ = load_pretrained(rp_, device, trainable=False, head='score', dtype=dt, model_max_length=mml, padding_side='right',
                                                                      build_kwargs=rpk)
            model_cfg = actor.cfg
        else:
            actor = build_model(model_cfg, device, trainable=True, dtype=dt, **epk)
            ref = build_model(model_cfg, device, trainable=False, dtype=dt, **epk)
            if actor_state is not None:
                actor.load_state_dict(actor_state)
            if reference_state is not None or actor_state is not None:
                ref.load_state_dict(reference_state if reference_state is not None else actor_state)
            if self.reward_fn is None:
                reward = build_model(reward_model_cfg or model_cfg, device, trainable=False, head='score', dtype=dt, **rpk)
                if reward_state is not None:
                    reward.load_state_dict(reward_state)
        self.model_cfg, self.reward_model_cfg = model_cfg, reward_model_cfg
        self._modules = {'actor': actor, 'ref': ref, 'reward': reward}

    def init_datasets(self) -> None:
        """grpo.py:135-139 `get_dataloaders(PromptOnlyDataset, PromptOnlyDataset, None)` through the reference's own dataset / template plugins."""
        if not self._from_paths:
            return               # injected models carry no tokenizer: the caller hands train() its dataloader
        from .common import get_dataloaders
        self.prompt_only_dataloader, self.eval_dataloader = get_dataloaders(self, 'PromptOnlyDataset', 'PromptOnlyDataset', rl=True)

    def init_engines(self) -> None:
        # grpo.py:153-181: the schedule length comes from the prompt dataloader that train() may still receive; until then it is unknown
        # (an explicit train_cfgs.total_training_steps wins) and a cosine engine refuses to step rather than decay to 0
        cfgs, ds_cfgs, mods = self.cfgs, self.ds_train_cfgs, self._modules
        t = lambda k, d: cfg_get(cfgs, 'train_cfgs.' + k, d)
        self.gas = int(cfg_get(ds_cfgs, 'gradient_accumulation_steps', t('gradient_accumulation_steps', 1)))
        total = t('total_training_steps', None)
        total = None if total is None else max(1, int(total) // self.gas)
        self.actor_model = NativeEngine(
            mods['actor'], lr=float(t('actor_lr', 1e-6)), betas=[float(b) for b in t('adam_betas', [0.9, 0.95])],
            weight_decay=float(t('actor_weight_decay', 0.01)), max_grad_norm=float(cfg_get(ds_cfgs, 'gradient_clipping', 1.0)),
            total_steps=total, warmup_steps=int(float(t('actor_lr_warmup_ratio', 0.03)) * (total or 0)),
            lr_scheduler_type=t('actor_lr_scheduler_type', 'cosine'), gradient_accumulation_steps=self.gas)
        self.actor_reference_model = NativeEngine(mods['ref'], trainable=False)
        self.reward_model = NativeEngine(mods['reward'], trainable=False) if mods['reward'] is not None else None
        self._modules = None

    def init_logger(self) -> None:
        self.logger = None          # observability is out of scope; train() returns the metrics

    def set_train(self, mode: bool = True) -> None:
        """base/rl_trainer.py:274-286."""
        self.actor_model.module.train(mode)

    # ------------------------------------------------------------------ grpo.py:212-227
    @ops.few_row_gemms
    def generate_completions(self, prompt_batch, generator=None):
        """`generate(num_return_sequences=G, do_sample=True)`: row b*G+g is sample g of prompt b (HF expands the
        batch with repeat_interleave before sampling)."""
        from ..generation import generate
        m = lambda k, d: cfg_get(self.cfgs, 'model_cfgs.' + k, d)
        G = self.num_generations
        self.actor_model.wait_optimizer()
        return generate(self.actor_model.module, prompt_batch['input_ids'].repeat_interleave(G, 0),
                        prompt_batch['attention_mask'].repeat_interleave(G, 0), max_length=int(m('model_max_length', 2048)),
                        do_sample=True, temperature=float(m('temperature', 1.0)), top_p=float(m('top_p', 1.0)), top_k=m('top_k', 'hf'),
                        repetition_penalty=float(m('repetition_penalty', 1.0)), eos_token_id=self.eos_token_id,
                        pad_token_id=self.pad_token_id, generator=generator)

    # ------------------------------------------------------------------ grpo.py:229-255
    def compute_rewards(self, sequences, prompt_length):
        completions = sequences[:, prompt_length:]
        keep = ops.completion_mask(completions, self.eos_token_id)
        if self.reward_fn is not None:
            return self.reward_fn(completions * keep.to(completions.dtype)).to(torch.float32)
        tok, rtok = getattr(self, 'tokenizer', None), getattr(self, 'reward_tokenizer', None)
        if tok is not None and rtok is not None:
            # grpo.py:243-250 `batch_retokenize`, as the reference does it whatever the tokenizers are: the masked ids (0 after the first eos) are DECODED
            # with the actor's tokenizer without special tokens, the reward tokenizer's eos is appended to every text (also to a completion that was
            # cut at the length cap) and the texts are ENCODED with the reward tokenizer (its bos, if it adds one; padded to the longest on its side).
            # Host string work, exactly the reference's; the id shortcut below is for trainers built from injected models, which carry no tokenizer.
            texts = tok.batch_decode(completions * keep.to(completions.dtype), skip_special_tokens=True)
            enc = rtok([t + rtok.eos_token for t in texts], padding=True, truncation=False, return_tensors='pt')
            ids, am = enc['input_ids'].to(sequences.device), enc['attention_mask'].to(sequences.device)
        else:
            # same-tokenizer shortcut: the kept tokens are the reward model's input, everything after the first EOS is padding
            ids = torch.where(keep.bool(), completions, torch.full_like(completions, self.pad_token_id))
            am = keep.to(torch.int64)
        scores = self.reward_model.module.scores(ids, am)
        end = end_index(self.reward_model.module.kind, am)
        return scores[torch.arange(ids.shape[0], device=ids.device), end].float()

    # ------------------------------------------------------------------ grpo.py:199-210
    def _get_per_token_logps(self, engine, input_ids, attention_mask, logits_to_keep, save=False):
        """log_softmax(logits[:, :-1][:, -K:]).gather(input_ids[:, -K:]) -> fp32 [N, K]: hidden position j in
        [T-1-K, T-2] predicts token j+1.  Only those N*K rows go through the lm_head."""
        w = build_span_window(input_ids, input_ids.shape[1] - 1 - logits_to_keep)
        if hasattr(engine, 'wait_optimizer'):
            engine.wait_optimizer()
        lp = engine.module.response_logprobs(input_ids, attention_mask, w, save=save)
        return lp[:w['rows']].view(w['N'], w['W']), w

    # ------------------------------------------------------------------ grpo.py:257-329
    @ops.few_row_gemms
    def train_step(self, prompt_batch, generator=None, sequences=None, rewards=None):
        """`sequences` / `rewards` may be injected (tests, external samplers or rule-based rewards); otherwise they come
        from `generate_completions` / `compute_rewards` as in the reference."""
        prompt_batch = {k: v.to(self.device) for k, v in prompt_batch.items()}
        P = prompt_batch['input_ids'].size(1)
        B, G = prompt_batch['input_ids'].size(0), self.num_generations
        if sequences is None:
            sequences = self.generate_completions(prompt_batch, generator)
        sequences = sequences.to(self.device).contiguous()
        if rewards is None:
            rewards = self.compute_rewards(sequences, P)
        rewards = rewards.to(self.device, torch.float32).contiguous()
        advantages = ops.group_advantage(rewards, B, G)

        attention_mask = (sequences != self.pad_token_id).long()
        K = sequences.size(1) - P
        ref_logps, _ = self._get_per_token_logps(self.actor_reference_model, sequences, attention_mask, K)
        logps, w = self._get_per_token_logps(self.actor_model, sequences, attention_mask, K, save=True)
        cmask = ops.completion_mask(sequences[:, P:], self.eos_token_id)
        loss, dlogp = ops.grpo_loss(logps.contiguous(), ref_logps.contiguous(), advantages, cmask, self.beta)

        self.actor_model.set_pending(pad_rows(dlogp, w['rows_pad']))
        self.actor_model.backward(loss)
        self.actor_model.step()
        s = get_all_reduce_mean(torch.stack([loss.reshape(()), rewards.mean()])).tolist()
        return {'train/loss': s[0], 'train/reward': s[1]}

    def train(self, prompt_only_dataloader=None, generator=None) -> list:
        """grpo.py:330-386 without its logging: one `train_step` (rollout of `num_generations` completions per prompt, rewards, update) per
        prompt batch and epoch; resumes at `self.global_step` (:347-357), saves `slice_<global_step>` every epochs * len(dataloader) //
        logger_cfgs.save_total_limit steps (:368-375) and -- when `logger_cfgs.output_dir` is configured -- the final model (:386; the reference
        always writes it, the native loop does not create ./output on its own).  Returns the per-step metrics."""
        if prompt_only_dataloader is None:      # the loader init_datasets() built from data_cfgs (the cfgs-only constructor)
            prompt_only_dataloader = getattr(self, 'prompt_only_dataloader', None)
            if prompt_only_dataloader is None:
                raise ValueError('GRPOTrainer.train needs a prompt dataloader (argument, or data_cfgs.train_datasets with the cfgs-only constructor)')
        history = []
        self.global_step = getattr(self, 'global_step', 0)
        t = lambda k, d: cfg_get(self.cfgs, 'train_cfgs.' + k, d)
        epochs = int(t('epochs', 1))
        n = len(prompt_only_dataloader) if hasattr(prompt_only_dataloader, '__len__') else None
        if self.actor_model.total_steps is None and n is not None:
            total = (n * epochs * int(t('update_iters', 1)) * int(t('per_device_prompt_batch_size', 1))
                     // max(1, int(t('per_device_train_batch_size', 1))))            # grpo.py:157-163
            self.actor_model.set_schedule(max(1, total // self.gas), float(t('actor_lr_warmup_ratio', 0.03)))
        remain = epochs - self.global_step // n if n else epochs
        skip = self.global_step % n if n else 0
        every = save_interval(self.cfgs, epochs * n if n else None)
        self.eval_history = getattr(self, 'eval_history', [])
        if eval_due(self.cfgs, 'begin'):                                          # grpo.py:345-346
            self.eval_history.append((self.global_step, self.eval()))
        for epoch in range(int(remain)):
            for i, batch in enumerate(prompt_only_dataloader):
                if epoch == 0 and i < skip:
                    continue
                history.append(self.train_step(batch, generator))
                self.global_step += 1
                if every and self.global_step % every == 0:
                    self.save(tag=self.global_step)
                if eval_due(self.cfgs, 'steps', self.global_step):                # grpo.py:378-384
                    self.eval_history.append((self.global_step, self.eval()))
        if cfg_get(self.cfgs, 'logger_cfgs.output_dir', None):
            self.save()
        return history

    @ops.few_row_gemms
    def actor_step(self, prompt_batch, generator=None):
        """One sampled completion per prompt (the evaluation loop's `generate`, base/rl_trainer.py:302-309)."""
        from ..generation import generate
        m = lambda k, d: cfg_get(self.cfgs, 'model_cfgs.' + k, d)
        self.actor_model.wait_optimizer()
        seq = generate(self.actor_model.module, prompt_batch['input_ids'], prompt_batch['attention_mask'], max_length=int(m('model_max_length', 2048)),
                       do_sample=True, temperature=float(m('temperature', 1.0)), top_p=float(m('top_p', 1.0)), top_k=m('top_k', 'hf'),
                       repetition_penalty=float(m('repetition_penalty', 1.0)), eos_token_id=self.eos_token_id, pad_token_id=self.pad_token_id,
                       generator=generator)
        return {'input_ids': seq, 'attention_mask': seq.ne(self.pad_token_id)}

    def eval(self, eval_dataloader=None) -> dict:
        """base/rl_trainer.py:289-329 (common.rl_eval): actor completions for the evaluation prompts."""
        return rl_eval(self, eval_dataloader)

    def save(self, model=None, tag=None, output_dir=None) -> str:
        """base/rl_trainer.py save_transformers: the ACTOR in the layout `from_pretrained` loads (common.save_slice)."""
        return save_slice(self, model or self.actor_model, tag, output_dir)
