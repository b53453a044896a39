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

    def build_kl_dataloader(self):
        """kto.py:50-69: the reference's own `UnmatchedSupervisedDataset` (prompt i with the response of sample i - 1) over
        `data_cfgs.train_datasets` with the training template, batches of `train_cfgs.per_device_kl_batch_size`, shuffled -- built when the
        trainer came from `model_cfgs.model_name_or_path` (it needs the tokenizer and the template `init_datasets()` made).  None otherwise:
        the caller hands `train()` its unmatched batches."""
        template = getattr(self, 'train_template', None)
        path = cfg_get(self.cfgs, 'data_cfgs.train_datasets', None)
        if self.tokenizer is None or template is None or not path or isinstance(template, (list, tuple)):
            return None
        import importlib
        import torch.distributed as dist
        from torch.utils.data import DataLoader
        from torch.utils.data.distributed import DistributedSampler
        from ..data import DevicePrefetcher
        from .common import infer_modality
        d = lambda k, default=None: cfg_get(self.cfgs, 'data_cfgs.' + k, default)
        mod = importlib.import_module(f'align_anything.datasets.{infer_modality(self)}.supervised')
        ds = mod.UnmatchedSupervisedDataset(path=path, template=template, tokenizer=self.tokenizer, processor=self.processor, name=d('train_name'), size=d('train_size'),
                                            split=d('train_split'), data_files=d('train_data_files'), optional_args=d('train_optional_args', []))
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        rank = dist.get_rank() if world > 1 else 0
        loader = DataLoader(ds, collate_fn=ds.get_collator(), sampler=DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True),
                            batch_size=int(cfg_get(self.cfgs, 'train_cfgs.per_device_kl_batch_size', 64)))
        return DevicePrefetcher(loader, self.device, self.pad_token_id)

    def train(self, kl_dataloader=None):
        """kto.py:196-250 = the DPO loop (evaluation, checkpoint and resume schedule included: DPOTrainer.train) with one addition: at the start
        of every epoch whose first step index is a multiple of `kl_steps`, the KL estimate is refreshed over the unmatched batches
        (`kl_dataloader`, or the reference's own UnmatchedSupervisedDataset built from data_cfgs: build_kl_dataloader) -- each batch overwrites
        the estimate, the last one stands, as in the reference's loop -- then the ordinary preference steps run."""
        self._kl_loader = kl_dataloader if kl_dataloader is not None else self.build_kl_dataloader()
        try:
            return super().train()
        finally:
            self._kl_loader = None

    def _epoch_begin(self) -> None:
        kl_steps = max(1, int(cfg_get(self.cfgs, 'train_cfgs.kl_steps', 1)))
        if getattr(self, '_kl_loader', None) is not None and self.global_step % kl_steps == 0:
            for b in self._kl_loader:
                self.compute_kl(b)
