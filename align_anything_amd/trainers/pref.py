"""Sibling preference trainers on the native DPO machinery: SimPO, ORPO, KTO.

Each reference class subclasses its DPOTrainer and overrides `loss` only (align_anything/trainers/text_to_text/
simpo.py:39-108, orpo.py:39-112, kto.py:48-160); so do these.  The window log-probs come from the same
`compute_log_probs` path; the reference's slicing of the padded window tensor by absolute positions
([diverge_index, end_index + 1), identical pairs skipped) is reproduced on the device by `aa_pair_slice_index`, and the
loss + metrics + d loss / d logp by `aa_pref_loss_fwd_bwd` (csrc/pref_losses.hip)."""
from __future__ import annotations

import torch

from .. import ops
from .common import cfg_get
from .dpo import DPOTrainer


class _SlicedPreferenceTrainer(DPOTrainer):
    kind = None

    def _params(self):
        return 0.0, 0.0, 0.0

    def loss(self, batch) -> dict[str, torch.Tensor]:
        w = self._window(batch)
        B = w['N'] // 2
        ref = None
        if self.uses_reference:
            ref = self._flat_log_probs(self.reference_model.module, batch, save=False)
        self.model.wait_optimizer()
        pol = self._flat_log_probs(self.model.module, batch, save=True)
        am = batch.get('attention_mask')
        if am is None:
            am = torch.ones_like(batch['input_ids'])
        lo, hi, ln, keep = ops.pair_slice_index(batch['input_ids'], am, w['seq_off'], B)
        p1, p2, p3 = self._params()
        out7, per, dlogp = ops.pref_loss(self.kind, pol, ref, lo, hi, ln, keep, B, self.scale_coeff, p1, p2, p3)
        self.model.set_pending(dlogp)
        # the reference stacks only the kept pairs; `kept` is the boolean selector into the per-pair tensors
        kept = keep.bool()
        return {'loss': out7[0], 'reward': per[2][kept], 'better_sample_reward': per[0][kept], 'worse_sample_reward': per[1][kept],
                'reward_accuracy': out7[1], 'reward_margin': per[3][kept], '_means': out7[:6], 'kept': kept}


class SimPOTrainer(_SlicedPreferenceTrainer):
    """simpo.py: length-normalised log-probs, no reference model in the loss, margin `gamma`."""
    kind, uses_reference = 'simpo', False

    def _params(self):
        return float(cfg_get(self.cfgs, 'train_cfgs.gamma', 1.4)), 0.0, 0.0    # default: configs/train/text_to_text/simpo.yaml:61


class ORPOTrainer(_SlicedPreferenceTrainer):
    """orpo.py: SFT term on the chosen row + scale_coeff * odds-ratio term, no reference model in the loss."""
    kind, uses_reference = 'orpo', False


class KTOTrainer(_SlicedPreferenceTrainer):
    """kto.py: per-row log-ratios against the reference, shifted by a batch KL estimate (`compute_kl`)."""
    kind, uses_reference = 'kto', True

    def __init__(self, *a, **k):
        self.kl = 0.0
        super().__init__(*a, **k)

    def compute_kl(self, batch):
        """kto.py:50-81 on one unmatched batch: kl = max(mean(log_probs - ref_log_probs), 0) over the padded window
        tensor.  Returns the device scalar and keeps its value for `loss` (one host read, outside the step)."""
        w = self._window(batch)
        ref = self._flat_log_probs(self.reference_model.module, batch, save=False)
        self.model.wait_optimizer()
        pol = self._flat_log_probs(self.model.module, batch, save=False)
        kl = ops.window_kl(pol, ref, w['rows'], float(w['N'] * max(w['max_len'], 1)))
        self.kl = float(kl.item())
        return kl

    def _params(self):
        return (float(cfg_get(self.cfgs, 'train_cfgs.scale_better', 1.0)), float(cfg_get(self.cfgs, 'train_cfgs.scale_worse', 1.0)),
                float(self.kl))

    def train(self, kl_dataloader=None):
        """kto.py:196-240: at the start of every epoch whose first step index is a multiple of `kl_steps`, the KL estimate is refreshed
        over the unmatched batches (`kl_dataloader`: prompts paired with a neighbour's response, datasets/text_to_text/supervised.py
        UnmatchedSupervisedDataset) -- each batch overwrites the estimate, the last one stands, as in the reference's loop -- then the
        ordinary preference steps run."""
        history = []
        epochs = int(cfg_get(self.cfgs, 'train_cfgs.epochs', 1))
        kl_steps = max(1, int(cfg_get(self.cfgs, 'train_cfgs.kl_steps', 1)))
        self.model.train()
        for _ in range(epochs):
            if kl_dataloader is not None and self.global_step % kl_steps == 0:
                for b in kl_dataloader:
                    self.compute_kl(b)
            for batch in self.train_dataloader:
                info = self.train_step(batch)
                self.global_step += 1
                info['train/epoch'] = self.global_step / max(1, len(self.train_dataloader))
                history.append(info)
            self.model.tput_timer.update_epoch_count()
        return history

