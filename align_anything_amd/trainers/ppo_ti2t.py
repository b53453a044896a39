"""Native text+image -> text PPO (align_anything/trainers/text_image_to_text/ppo.py): the multimodal variant of the PPO
loop, BASELINE configs[2] (Qwen2-VL actor / reference / reward / critic).

It differs from the text-only trainer in how experience is laid out: generated sequences are rotated so ALL padding is
on the left (`move_padding_left`, :56-86), every tensor of the update is a per-row RESPONSE WINDOW of R_i = number of
generated tokens (`logits[idx, :-1][-R:]`, `scores[:, :-1][idx][-R:]`, :233-241, :302-325) right-padded to max(R), the
mask is `log_probs != 0`, and GAE starts at 0.  The image tensors of the prompt batch ride along to every forward."""
from __future__ import annotations

import torch

from .. import ops
from .common import build_tail_window, flat_to_padded, get_all_reduce_max, get_all_reduce_mean
from .ppo import PPOTrainer

MM_KEYS = ('pixel_values', 'image_grid_thw', 'position_ids3')


class PPOTrainerTI2T(PPOTrainer):
    @staticmethod
    def _mm(batch):
        return {k: batch[k] for k in MM_KEYS if k in batch and batch[k] is not None}

    def _rollout_micro_batch(self, per_device_train_batch_size: int) -> int:
        """text_image_to_text/ppo.py:206-269 rolls out the WHOLE prompt batch in one go (no micro-batch split): the processor's
        `pixel_values` are [sum of patches, C] with `image_grid_thw` describing all images, so row-slicing them would corrupt the
        tower input."""
        return 0

    def _pad_id(self):
        return self._token_id('pad_token_id', 0)

    # ------------------------------------------------------------------ ppo.py:174-205
    def finish_sequences(self, prompt_batch, sequences):
        """What actor_step does after `generate`: move the padding left, rebuild the mask, count the generated tokens of
        every row (non-pad tokens of the sequence minus non-pad tokens of its prompt)."""
        pad = self._pad_id()
        seq = ops.move_padding_left(sequences, pad)
        am = seq.ne(pad)
        lens = (am.sum(1) - prompt_batch['input_ids'].ne(pad).sum(1)).tolist()      # one host read per rollout
        return dict(self._mm(prompt_batch), input_ids=seq, attention_mask=am.to(torch.int64)), [int(x) for x in lens]

    @ops.few_row_gemms
    def actor_step(self, prompt_batch, generator=None):
        from ..generation import generate
        from .common import cfg_get
        m = lambda k, d: cfg_get(self.cfgs, 'model_cfgs.' + k, d)
        self.actor_model.wait_optimizer()
        mm = self._mm(prompt_batch)
        pv = mm.pop('pixel_values', None)
        mm.pop('position_ids3', None)
        T = prompt_batch['input_ids'].shape[1]
        seq = generate(self.actor_model.module, prompt_batch['input_ids'], prompt_batch['attention_mask'],
                       max_new_tokens=int(m('max_new_tokens', 512)), do_sample=True, temperature=float(m('temperature', 1.0)),
                       top_p=float(m('top_p', 1.0)), top_k=m('top_k', 'hf'), repetition_penalty=float(m('repetition_penalty', 1.0)),
                       eos_token_id=self._token_id('eos_token_id', None), pad_token_id=self._pad_id(), pixel_values=pv, generator=generator, **mm)
        return self.finish_sequences(prompt_batch, seq)

    # ------------------------------------------------------------------ ppo.py:206-269
    def _window_values(self, engine, batch, w, save=False, scores=False):
        mod = engine.module
        if hasattr(engine, 'wait_optimizer'):
            engine.wait_optimizer()
        mm = self._mm(batch)
        fn = mod.response_scores if scores else mod.response_logprobs
        flat = fn(batch['input_ids'], batch['attention_mask'], w, pixel_values=mm.pop('pixel_values', None), save=save, **mm)
        return flat_to_padded(flat, w)

    @ops.few_row_gemms
    def rollout(self, prompt_batch, generator=None, sequences=None):
        actor_batch, response_lens = (self.actor_step(prompt_batch, generator) if sequences is None
                                      else self.finish_sequences(prompt_batch, sequences))
        ids, am = actor_batch['input_ids'], actor_batch['attention_mask']
        w = build_tail_window(ids, response_lens)
        mm = self._mm(actor_batch)
        pv = mm.pop('pixel_values', None)
        rs = self.reward_model.module.scores(ids, am, pv, **mm)
        T = ids.shape[1]
        end = (am * torch.arange(T, device=ids.device)[None]).argmax(dim=1)
        reward = rs[torch.arange(ids.shape[0], device=ids.device), end]           # end_scores (models/qwen2_vl.py:57-64)
        log_probs = self._window_values(self.actor_model, actor_batch, w)
        training = {'response_lens': response_lens, 'log_probs': log_probs,
                    'ref_log_probs': self._window_values(self.actor_reference_model, actor_batch, w), 'reward': reward,
                    'reward_values': self._window_values(self.reward_critic_model, actor_batch, w, scores=True),
                    'response_mask': log_probs != 0}
        return actor_batch, training

    # ------------------------------------------------------------------ ppo.py:271-379
    @ops.few_row_gemms
    def rl_step(self, inference_batch, training_batch):
        old_log_probs = training_batch['log_probs'].float().contiguous()
        ref_log_probs = training_batch['ref_log_probs'].float().contiguous()
        reward = training_batch['reward'].float().contiguous()
        old_values = training_batch['reward_values'].float().contiguous()
        mask = training_batch['response_mask'].bool().contiguous()
        w = build_tail_window(inference_batch['input_ids'], training_batch['response_lens'])
        m8 = self._m(mask)
        old_rewards, _ = ops.kl_reward(reward, old_log_probs, ref_log_probs, m8, self.kl_coeff, self.clip_range_score)
        adv, ret = ops.gae(old_values, old_rewards, m8, 0, self.gamma, self.gae_lambda)

        def to_flat(d):          # [B, maxR] gradient -> flat window rows (zero tail)
            out = torch.zeros(w['rows_pad'], dtype=torch.float32, device=d.device)
            out[:w['rows']] = d.reshape(-1)[w['flat_to_padded']]
            return out

        log_probs = self._window_values(self.actor_model, inference_batch, w, save=True)
        actor_loss, dlogp = ops.ppo_actor_loss(log_probs.contiguous(), old_log_probs, adv, m8, self.clip_range_ratio)
        self.actor_model.set_pending(to_flat(dlogp))
        self.actor_model.backward(actor_loss)
        self.actor_model.step()

        values = self._window_values(self.reward_critic_model, inference_batch, w, save=True, scores=True)
        critic_loss, dvalues = ops.ppo_critic_loss(values.contiguous(), old_values, ret, m8, self.clip_range_value)
        self.reward_critic_model.set_pending(to_flat(dvalues))
        self.reward_critic_model.backward(critic_loss)
        self.reward_critic_model.step()

        mf = mask.float()
        cnt = mf.sum(-1)
        mm_ = lambda x: ((x * mf).sum(-1) / cnt).mean()
        stats = torch.stack([actor_loss.reshape(()), critic_loss.reshape(()), reward.mean(), (old_rewards * mf).sum(-1).mean(), mm_(adv), mm_(ret),
                             mm_(values), ((old_log_probs - ref_log_probs) * mf).sum(-1).mean(), cnt.mean()])
        s = get_all_reduce_mean(stats).tolist()
        mx = get_all_reduce_max(cnt.max().reshape(1))
        return {'train/actor_loss': s[0], 'train/reward_critic_loss': s[1], 'train/reward': s[2], 'train/reward_with_kl_penalty': s[3],
                'train/reward_advantage': s[4], 'train/reward_return': s[5], 'train/reward_value': s[6], 'train/kl_divergence': s[7],
                'train/actor_lr': self.actor_model.optimizer.param_groups[0]['lr'],
                'train/reward_critic_lr': self.reward_critic_model.optimizer.param_groups[0]['lr'],
                'train/mean_generated_length': s[8], 'train/max_generated_length': float(mx.item())}
