"""Native supervised fine-tuning step: align_anything/trainers/text_to_text/sft.py:94-108 (`loss` = the HF causal-LM loss of
`model(**batch)` with the collator's labels, `train_step` = backward + step) on the DPO machinery -- the first stage of the
reference's pipeline (scripts/opt/sft.sh) behind the same boundary.

The HF loss (hf:loss/loss_utils.py ForCausalLMLoss) is the mean negative log-likelihood of labels[:, 1:] under logits[:, :-1]
over the positions whose label is not -100.  Here those positions are the rows of a window (`build_label_window`), the
log-probs come from the same fused norm -> lm_head -> log-softmax-gather path as DPO (lm_head only on the rows that carry a
label), and `aa_sft_loss_fwd_bwd` yields the loss and d loss / d logp in one launch."""
from __future__ import annotations

import torch

from .. import ops
from .common import build_label_window
from .dpo import DPOTrainer


class SupervisedTrainer(DPOTrainer):
    uses_reference = False
    dataset_types = ('SupervisedDataset', 'SupervisedDataset')          # sft.py:85-89

    def loss(self, sft_batch) -> dict[str, torch.Tensor]:
        w = sft_batch.get('_window') or build_label_window(sft_batch['labels'], device=sft_batch['input_ids'].device)
        self.model.wait_optimizer()
        mm = {k: sft_batch[k] for k in ('image_grid_thw', 'position_ids3', 'input_features', 'feature_attention_mask') if k in sft_batch}
        logp = self.model.module.response_logprobs(sft_batch['input_ids'], sft_batch.get('attention_mask'), w,
                                                   pixel_values=sft_batch.get('pixel_values'), save=True, **mm)
        loss, dlogp = ops.sft_loss(logp, w['rows'])
        self.model.set_pending(dlogp)
        return {'loss': loss}

    def train_step(self, sft_batch) -> dict:
        loss = self.loss(sft_batch)['loss']
        self.model.backward(loss)
        self.model.step()
        return {'train/loss': float(loss.item()), 'train/lr': self.model.optimizer.param_groups[0]['lr']}
