"""TEST INFRASTRUCTURE ONLY.  One DPO optimizer step of BASELINE.json configs[0] (OPT-125m, fp32) as a pure function of
(weights, Adam moments, step index, batch) -- the instrument that makes "loss curves matching the reference to 1e-4"
well posed at EVERY one of the 64 steps.

A free-running curve compares two chaotic trajectories: the reference cannot reproduce its own curve to 1e-4 when only
its CPU thread count changes (tests/golden/opt125m_curve.npz, `metrics_3threads`).  Teacher forcing removes the
trajectory from the comparison: step k starts from the SAME weights and moments on both sides, so the loss of step k
and the parameter update of step k are compared as functions, with no accumulated divergence.

Restates, with the oracle's own model port (oracle/models.py::opt_logits, pinned to reference fixtures by
tests/test_oracle_golden.py):
  * trainers/text_to_text/dpo.py:205-237  train_step = loss -> engine.backward -> engine.step
  * trainers/text_to_text/dpo.py:122-203  compute_log_probs + loss (oracle/rl_math.py)
  * trainers/base/supervised_trainer.py:234-257  FusedAdam(adam_w_mode) over utils/tools.py:241-270's two groups
    (no decay for names containing bias / *norm.weight), betas (0.9, 0.95), eps 1e-8, lr 1e-6, weight decay 0.05,
    `get_scheduler('cosine')` with int(0.03 x steps) warm-up steps, gradient_clipping 1.0 (global L2, torch form)

Pinned to the reference in the build container by oracle/gen_golden.py::gen_opt125m_teacher: the UNMODIFIED reference
trainer runs the 64 steps; before every step its weights and torch.optim.AdamW state are handed to `Teacher.step`, whose
loss / gradient norm / updated weights are compared with what the reference itself then produces (the maxima are stored
in tests/golden/opt125m_teacher.npz and asserted at generation time); tests/test_oracle_golden.py re-checks the first
steps against the stored reference numbers without the reference.
"""
from __future__ import annotations

import math

import torch

from . import models as om
from . import rl_math as orl

NO_DECAY = ('bias', 'layer_norm.weight', 'layernorm.weight', 'norm.weight', 'ln_f.weight')     # utils/tools.py:246-252


def cosine_lr(step: int, base_lr: float, warmup: int, total: int) -> float:
    """transformers.get_scheduler('cosine') (optimization.py get_cosine_schedule_with_warmup): lr used BY update `step` (0-based)."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    prog = (step - warmup) / max(1, total - warmup)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


class Teacher:
    def __init__(self, cfg: dict, ref_sd: dict, pad_token_id: int, total_steps: int, *, beta=0.1, lr=1e-6, betas=(0.9, 0.95), eps=1e-8,
                 weight_decay=0.05, warmup_ratio=0.03, max_grad_norm=1.0, hf_config=None, device=None):
        """hf_config (an OPTConfig): evaluate the model with HuggingFace's own OPTForCausalLM (`torch.func.functional_call` on the given
        weights) -- the arithmetic the reference itself executes (models/opt.py:28 subclasses it) -- instead of the
        oracle's port (oracle/models.py::opt_logits, whose fp32 operation order differs: ~2e-5 on the loss at identical weights)."""
        # device ('cuda:0' in the GPU test): the SAME code -- HF modules, torch autograd, torch ops -- executed by torch on the GPU in fp32 (eager
        # attention), 60 x faster than the host cores of the GPU box; every tensor handed to step() must then live there.  The CPU form is the one
        # pinned bit for bit to the reference; the GPU test ties the two together at a few steps.
        self.device = torch.device(device) if device is not None else torch.device('cpu')
        self.cfg, self.ref_sd, self.pad = cfg, {k: v.detach().to(self.device) for k, v in ref_sd.items()}, pad_token_id
        self.total, self.warmup = int(total_steps), int(warmup_ratio * total_steps)
        self.beta, self.lr, self.betas, self.eps, self.wd, self.max_norm = beta, lr, betas, eps, weight_decay, max_grad_norm
        self.hf = None
        if hf_config is not None:
            import copy
            from transformers import OPTForCausalLM
            hc = copy.deepcopy(hf_config)
            if self.device.type != 'cpu':
                hc._attn_implementation = 'eager'          # plain matmul / softmax in fp32 (no dependence on which fused SDPA kernels the build has)
            self.hf = OPTForCausalLM(hc).eval().to(self.device)

    def logits(self, sd, ids, am):
        if self.hf is None:
            return om.opt_logits(sd, self.cfg, ids, am)
        params = {k: v for k, v in sd.items() if k != 'lm_head.weight'}          # tied: functional_call re-ties it to the embedding
        return torch.func.functional_call(self.hf, params, (ids,), {'attention_mask': am}).logits

    # parameters whose TRUE gradient is identically zero: softmax is invariant to adding a constant to every score of a query row, and
    # q . b_k is exactly that -- what autograd returns for a key bias is rounding noise (~1e-9 against ~1e-3 elsewhere), which Adam
    # normalises to steps of +-lr in a random direction.  No two fp32 implementations agree on it (the reference does not agree with itself
    # across thread counts); their updates are bounded by 2 lr and excluded from the relative-error statistics.
    NOISE_ONLY = ('k_proj.bias',)

    @staticmethod
    def names(sd):
        """Trainable tensors: the tied lm_head is the embedding (hf OPTForCausalLM ties them; one parameter, one update)."""
        return [n for n in sd if n != 'lm_head.weight']

    def step(self, w: dict, m: dict, v: dict, k: int, batch: dict):
        """One optimizer update, 0-based index k, from weights `w` and moments `m`, `v` (dicts by HF name; not modified).
        Returns (info, w_next, m_next, v_next); info = loss, the other train/* metrics, grad_norm (pre-clip), lr used."""
        names = self.names(w)
        leaf = {n: w[n].detach().clone().requires_grad_(True) for n in names}
        sd = dict(leaf)
        sd['lm_head.weight'] = leaf['model.decoder.embed_tokens.weight']
        ids, am, lens = batch['input_ids'].to(self.device), batch['attention_mask'].to(self.device), batch['meta_info']['response_lens']
        logits = self.logits(sd, ids, am)
        lp = orl.compute_log_probs(logits, ids, lens, self.pad)
        with torch.no_grad():
            ref_lp = orl.compute_log_probs(self.logits(self.ref_sd, ids, am), ids, lens, self.pad)
        out = orl.dpo_loss(lp, ref_lp, self.beta)
        out['loss'].backward()
        grads = {n: leaf[n].grad for n in names}
        # torch.nn.utils.clip_grad_norm_ in its own operation order (norm of the per-tensor norms), so that on the reference's weights the
        # teacher reproduces the reference's update bit for bit where the model arithmetic is HF's (hf_config given)
        total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, 2.0) for g in grads.values()]), 2.0)
        coef = torch.clamp(self.max_norm / (total + 1e-6), max=1.0)
        lr = cosine_lr(k, self.lr, self.warmup, self.total)
        w2, m2, v2 = {}, {}, {}
        for n in names:
            p, mm, vv = w[n].detach().clone(), m[n].clone(), v[n].clone()
            wd = 0.0 if any(nd in n for nd in NO_DECAY) else self.wd
            # torch.optim.AdamW form (what oracle/gen_golden.py drives the reference with): decoupled decay first, then the Adam step
            p.mul_(1.0 - lr * wd)
            g = grads[n] * coef
            mm.lerp_(g, 1.0 - self.betas[0])
            vv.mul_(self.betas[1]).addcmul_(g, g, value=1.0 - self.betas[1])
            bc1, bc2 = 1.0 - self.betas[0] ** (k + 1), 1.0 - self.betas[1] ** (k + 1)
            denom = (vv.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(mm, denom, value=-lr / bc1)
            w2[n], m2[n], v2[n] = p, mm, vv
        w2['lm_head.weight'] = w2['model.decoder.embed_tokens.weight']
        info = {'train/loss': float(out['loss'].detach()), 'train/reward': float(out['reward'].mean()),
                'train/better_sample_reward': float(out['better_sample_reward'].mean()),
                'train/worse_sample_reward': float(out['worse_sample_reward'].mean()),
                'train/reward_accuracy': float(out['reward_accuracy']), 'train/reward_margin': float(out['reward_margin'].mean()),
                'train/lr': cosine_lr(k + 1, self.lr, self.warmup, self.total), 'grad_norm': float(total), 'lr_used': lr}
        return info, w2, m2, v2

    @staticmethod
    def zeros_like(w):
        return {n: torch.zeros_like(w[n]) for n in Teacher.names(w)}


def fingerprint_index(sd: dict, per_tensor: int = 16, seed: int = 99) -> dict:
    """A fixed sample of element positions per tensor (seeded): the committed fingerprint of the reference's weights after every step."""
    g = torch.Generator().manual_seed(seed)
    return {n: torch.randint(0, sd[n].numel(), (per_tensor,), generator=g) for n in sorted(Teacher.names(sd))}


def fingerprint(sd: dict, index: dict) -> torch.Tensor:
    return torch.stack([sd[n].detach().reshape(-1)[i] for n, i in index.items()]).to(torch.float32)
