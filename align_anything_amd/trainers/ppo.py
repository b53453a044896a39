"""PPO inner-loop math with the reference's method signatures, executed by the HIP kernels.

Mirrors align_anything/trainers/text_to_text/ppo.py: add_kl_divergence_regularization (:528-547),
get_advantages_and_returns (:487-508), actor_loss_fn (:291-307), critic_loss_fn (:510-526) and
utils/tools.py masked_mean (:460-467).  Each loss returns (loss, grad) -- the fused kernels emit d loss / d input,
which `NativeEngine.backward` consumes the same way the DPO step does (set_pending).  The python loop over t of the
reference's GAE (4*L kernel launches) is one launch here; the per-row `.nonzero()` host syncs of the KL reward are
a device-side last-index scan.

Round-1 scope: the math + actor log-prob path.  Rollout (`generate`), reward/critic score-head models and the
four-engine rl_step are the §8(f) "next" rows.
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
