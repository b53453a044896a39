"""TEST INFRASTRUCTURE ONLY.  fp32 CPU restatement of the reference's RLHF scalar math.

Every function cites the reference lines it follows (paths relative to /root/reference/align_anything).
Pinned by tests/test_oracle_golden.py against tests/golden/rl_math.npz, which oracle/gen_golden.py
produced by calling the reference's own unmodified functions.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def gather_log_probabilities(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """utils/tools.py:402-413 -- log_softmax over the vocab, then gather the label column."""
    log_probs = F.log_softmax(logits, dim=-1)
    return torch.gather(log_probs, dim=-1, index=labels.unsqueeze(-1).to(torch.int64)).squeeze(-1)


def strip_pad(seq: torch.Tensor, pad_token_id: int) -> torch.Tensor:
    """utils/tools.py strip_pad: drop every pad token of a 1-D id sequence."""
    return seq[seq != pad_token_id]


def response_window(input_ids: torch.Tensor, pad_token_id: int, response_len: int, T: int):
    """trainers/text_to_text/dpo.py:131-139 (same text_image_to_text/dpo.py:94-102).

    Returns (logit_positions, labels): the reference takes logits[idx][-R:][:-1] (positions T-R .. T-2)
    and labels strip_pad(ids)[-R:][1:].  Integer work -> must match bit-exactly.
    """
    raw = strip_pad(input_ids, pad_token_id)
    labels = raw[-response_len:][1:]
    pos = torch.arange(T - response_len, T - 1)
    return pos, labels


def compute_log_probs(logits: torch.Tensor, input_ids: torch.Tensor, response_lens, pad_token_id: int):
    """trainers/text_to_text/dpo.py:122-142: per-row window log-probs, right-padded with 0.0."""
    out = []
    for idx in range(len(response_lens)):
        R = response_lens[idx]
        raw = strip_pad(input_ids[idx], pad_token_id)
        logit = logits[idx][-R:].unsqueeze(0)
        ids = raw[-R:].unsqueeze(0)
        out.append(gather_log_probabilities(logit[:, :-1], ids[:, 1:]).squeeze(0))
    return torch.nn.utils.rnn.pad_sequence(out, batch_first=True, padding_value=0.0)


def dpo_loss(seq_logp: torch.Tensor, ref_seq_logp: torch.Tensor, beta: float) -> dict:
    """trainers/text_to_text/dpo.py:144-203 given the padded [2B, L-1] log-prob tensors."""
    better, worse = seq_logp.chunk(2, dim=0)
    ref_better, ref_worse = ref_seq_logp.chunk(2, dim=0)
    losses, br, wr = [], [], []
    for i in range(better.size(0)):
        b = better[i].sum(-1); w = worse[i].sum(-1)
        rb = ref_better[i].sum(-1); rw = ref_worse[i].sum(-1)
        blr, wlr = b - rb, w - rw
        losses.append(-F.logsigmoid(beta * (blr - wlr)))
        br.append(beta * blr.detach()); wr.append(beta * wlr.detach())
    loss = torch.stack(losses).mean()
    br = torch.stack(br); wr = torch.stack(wr)
    return {
        'loss': loss,
        'reward': br + wr,
        'better_sample_reward': br,
        'worse_sample_reward': wr,
        'reward_accuracy': (br > wr).float().mean(),
        'reward_margin': br - wr,
    }


def masked_mean(x: torch.Tensor, mask=None) -> torch.Tensor:
    """utils/tools.py:460-467."""
    if mask is None:
        return x.mean()
    return ((x * mask).sum(dim=-1) / mask.sum(dim=-1)).mean()


def add_kl_divergence_regularization(reward, log_probs, ref_log_probs, sequence_mask, kl_coeff, clip_range_score):
    """trainers/text_to_text/ppo.py:528-547."""
    end_index = torch.cat([m.nonzero()[-1] for m in sequence_mask])
    kl = log_probs - ref_log_probs
    pen = -kl_coeff * kl
    rewards = torch.scatter_add(pen, dim=-1, index=end_index.unsqueeze(-1), src=reward.to(pen.dtype).unsqueeze(-1))
    return torch.clamp(rewards, min=-clip_range_score, max=clip_range_score)


def get_advantages_and_returns(values, rewards, sequence_mask, start, gamma, gae_lambda):
    """trainers/text_to_text/ppo.py:487-508 (GAE reverse scan)."""
    last = 0.0
    adv_rev = []
    values = values * sequence_mask
    rewards = rewards * sequence_mask
    length = rewards.size(-1)
    for t in reversed(range(start, length)):
        nxt = values[:, t + 1] if t < length - 1 else 0.0
        delta = rewards[:, t] + gamma * nxt - values[:, t]
        last = delta + gamma * gae_lambda * last
        adv_rev.append(last)
    adv = torch.stack(adv_rev[::-1], dim=1)
    return adv.detach(), adv + values[:, start:]


def actor_loss_fn(log_probs, old_log_probs, advantages, mask, clip_range_ratio):
    """trainers/text_to_text/ppo.py:291-307."""
    ratios = torch.exp(log_probs - old_log_probs)
    s1 = advantages * ratios
    s2 = advantages * torch.clamp(ratios, 1.0 - clip_range_ratio, 1.0 + clip_range_ratio)
    return -masked_mean(torch.minimum(s1, s2), mask)


def critic_loss_fn(values, old_values, returns, mask, clip_range_value):
    """trainers/text_to_text/ppo.py:510-526."""
    vc = torch.clamp(values, old_values - clip_range_value, old_values + clip_range_value)
    l1 = torch.square(values - returns)
    l2 = torch.square(vc - returns)
    return 0.5 * masked_mean(torch.maximum(l1, l2), mask)


def adamw_step(p, g, m, v, step, lr, beta1, beta2, eps, wd):
    """DeepSpeed FusedAdam (adam_w_mode=True, bias_correction=True; third-party, multi_tensor_adam.cu
    ADAM_MODE_1) as used at trainers/base/supervised_trainer.py:245-249, on fp32 masters.
    Algebraically identical to torch.optim.AdamW up to rounding."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v / bc2).sqrt() + eps
    p.add_((m / bc1) / denom + wd * p, alpha=-lr)
    return p, m, v


def clip_coef(grads, max_norm: float):
    """configs/deepspeed/ds_z3_config.json:21 gradient_clipping (global L2), torch clip_grad_norm_ form."""
    total = torch.sqrt(sum((g.float() ** 2).sum() for g in grads))
    return torch.clamp(max_norm / (total + 1e-6), max=1.0), total


def grpo_loss(per_token_logps, ref_per_token_logps, rewards, B, G, completion_tokens, eos_token_id, beta):
    """trainers/text_to_text/grpo.py:270-316: group-normalised advantages (torch.std is unbiased), per-token k3 KL,
    loss = -(exp(logp - logp.detach()) * A - beta * KL), masked mean over tokens up to and including the first EOS."""
    r = rewards.view(B, G)
    adv = ((r - r.mean(dim=1, keepdim=True)) / (r.std(dim=1, keepdim=True) + 1e-4)).view(-1, 1)
    kl = torch.exp(ref_per_token_logps - per_token_logps) - (ref_per_token_logps - per_token_logps) - 1
    L = per_token_logps.size(1)
    per_token_loss = -(torch.exp(per_token_logps - per_token_logps.detach()) * adv.expand(-1, L) - beta * kl)
    mask = torch.ones_like(completion_tokens)
    for i in range(completion_tokens.size(0)):
        pos = (completion_tokens[i] == eos_token_id).nonzero(as_tuple=False)
        if pos.numel() > 0:
            mask[i, pos[0].item() + 1:] = 0
    mask = mask.to(per_token_loss.dtype)
    return (per_token_loss * mask).sum() / mask.sum(), adv.view(-1), mask


def _pair_slices(better_ids, worse_ids, better_mask, worse_mask, i):
    """simpo.py:64-77 (identical in orpo.py / kto.py): None for identical rows, else (better_slice, worse_slice,
    better_length, worse_length) in ABSOLUTE positions -- applied by the callers to the padded window tensor."""
    if torch.all(torch.eq(better_ids[i], worse_ids[i])).item():
        return None
    be = better_mask[i].nonzero()[-1].squeeze().item()
    we = worse_mask[i].nonzero()[-1].squeeze().item()
    d = (better_ids[i] != worse_ids[i]).nonzero()[0].squeeze().item()
    return slice(d, be + 1), slice(d, we + 1), be + 1, we + 1


def _pref_metrics(losses, br, wr):
    br, wr = torch.stack(br), torch.stack(wr)
    return {'loss': torch.stack(losses).mean(), 'reward': br + wr, 'better_sample_reward': br, 'worse_sample_reward': wr,
            'reward_accuracy': (br > wr).float().mean(), 'reward_margin': br - wr}


def simpo_loss(seq_logp, input_ids, attention_mask, scale_coeff, gamma):
    """trainers/text_to_text/simpo.py:41-108 given the padded [2B, L-1] window log-probs."""
    b, w = seq_logp.chunk(2, dim=0)
    bi, wi = input_ids.chunk(2, dim=0)
    bm, wm = attention_mask.chunk(2, dim=0)
    losses, brs, wrs = [], [], []
    for i in range(bi.size(0)):
        sl = _pair_slices(bi, wi, bm, wm, i)
        if sl is None:
            continue
        blr = b[i, sl[0]].sum(-1) / sl[2]
        wlr = w[i, sl[1]].sum(-1) / sl[3]
        losses.append(-F.logsigmoid(scale_coeff * (blr - wlr) - gamma))
        brs.append(scale_coeff * blr.detach()); wrs.append(scale_coeff * wlr.detach())
    return _pref_metrics(losses, brs, wrs)


def orpo_loss(seq_logp, input_ids, attention_mask, scale_coeff):
    """trainers/text_to_text/orpo.py:41-112."""
    b, w = seq_logp.chunk(2, dim=0)
    bi, wi = input_ids.chunk(2, dim=0)
    bm, wm = attention_mask.chunk(2, dim=0)
    losses, brs, wrs = [], [], []
    for i in range(bi.size(0)):
        sl = _pair_slices(bi, wi, bm, wm, i)
        if sl is None:
            continue
        blr = b[i, sl[0]].sum(-1) / sl[2]
        wlr = w[i, sl[1]].sum(-1) / sl[3]
        log_odds = (blr - wlr) - (torch.log1p(-torch.exp(blr)) - torch.log1p(-torch.exp(wlr)))
        losses.append(-blr + scale_coeff * -F.logsigmoid(log_odds))
        brs.append(scale_coeff * blr.detach()); wrs.append(scale_coeff * wlr.detach())
    return _pref_metrics(losses, brs, wrs)


def kto_loss(seq_logp, ref_seq_logp, input_ids, attention_mask, scale_coeff, scale_better, scale_worse, kl):
    """trainers/text_to_text/kto.py:83-160 (the minus in front of the worse term is the reference's)."""
    b, w = seq_logp.chunk(2, dim=0)
    rb, rw = ref_seq_logp.chunk(2, dim=0)
    bi, wi = input_ids.chunk(2, dim=0)
    bm, wm = attention_mask.chunk(2, dim=0)
    losses, brs, wrs = [], [], []
    for i in range(bi.size(0)):
        sl = _pair_slices(bi, wi, bm, wm, i)
        if sl is None:
            continue
        blr = b[i, sl[0]].sum(-1) - rb[i, sl[0]].sum(-1)
        wlr = w[i, sl[1]].sum(-1) - rw[i, sl[1]].sum(-1)
        losses.append(scale_better * (1 - torch.sigmoid(scale_coeff * (blr - kl)))
                      - scale_worse * (1 - torch.sigmoid(scale_coeff * (kl - wlr))))
        brs.append(scale_coeff * blr.detach()); wrs.append(scale_coeff * wlr.detach())
    return _pref_metrics(losses, brs, wrs)


def kto_kl(seq_logp, ref_seq_logp):
    """kto.py:74-81: mean over the padded tensors, clamped at 0."""
    return torch.clamp((seq_logp - ref_seq_logp).mean(), min=0.0)


def sft_loss(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
    """trainers/text_to_text/sft.py:94-97 `outputs.loss` = hf:loss/loss_utils.py ForCausalLMLoss: logits upcast to fp32, labels shifted
    left by one (the last position predicts nothing), mean cross-entropy over the positions whose label is not `ignore_index`."""
    logits = logits.float()
    shift = F.pad(labels, (0, 1), value=ignore_index)[..., 1:]
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), shift.reshape(-1), ignore_index=ignore_index, reduction='mean')


def rm_loss(scores: torch.Tensor, end_scores: torch.Tensor, regularization: float) -> dict:
    """align_anything/trainers/text_to_text/rm.py:97-132 -- scores [2B, L, 1] and end_scores [2B, 1] of the score model
    (rows [0, B) = better, [B, 2B) = worse): pairwise -logsigmoid, optional L2 on the end rewards, all six outputs."""
    higher_rewards, lower_rewards = scores.squeeze(dim=-1).chunk(chunks=2, dim=0)
    higher_end, lower_end = end_scores.squeeze(dim=-1).chunk(chunks=2, dim=0)
    loss = -F.logsigmoid(higher_end - lower_end).mean()
    if regularization > 0.0:
        loss = loss + regularization * torch.stack([lower_end, higher_end]).square().mean()
    return {'loss': loss, 'higher_end_reward': higher_end, 'lower_end_reward': lower_end, 'higher_rewards': higher_rewards,
            'lower_rewards': lower_rewards, 'accuracy': (higher_end > lower_end).float().mean()}
