"""Native reward-model trainer step (align_anything/trainers/text_to_text/rm.py:97-147): score-head model forward,
pairwise -logsigmoid(r+ - r-) + L2 regularisation on the end scores, backward, step.  Batches are RIGHT padded
(rm.py:81); the end score is the score at the last attended token for the text backbones (models/opt.py:67-89) and at
position -1 for the vision-language reward models (models/llava.py:64-68, qwen2_vl.py:61-64) -- common.end_index."""
from __future__ import annotations

import torch

from .. import ops
from ..engine import NativeEngine
from ..modeling import build_model
from .common import END_AT_LAST_POSITION_KINDS, cfg_get, eval_due, compute_dtype, end_index, get_all_reduce_mean, pad64, save_interval, save_slice


class RMTrainer:
    dataset_types = ('PreferenceDataset', 'PreferenceDataset')          # rm.py:87-91

    def __init__(self, cfgs, ds_cfgs=None, *, model_cfg=None, state=None, device='cuda:0', train_dataloader=None):
        """`RMTrainer(cfgs, ds_cfgs)` alone follows the reference's constructor (text_to_text/rm.py:52-72): the score model, tokenizer and
        processor come from `model_cfgs.model_name_or_path` (rm.py:76-85: `is_reward_model=True`, right padding; a language-model checkpoint
        without a score head starts with the native initialisation of it) and the dataloaders from `data_cfgs` (rm.py:87-91); the keyword
        arguments inject pre-built pieces instead.  The phases are the reference's own methods, in its order (rm.py:57-67), so a subclass
        overrides `init_models` / `init_datasets` / `loss` as the reference's modality trainers do."""
        self.cfgs, self.ds_train_cfgs, self.device = cfgs, ds_cfgs, torch.device(device)
        self.model_cfg = model_cfg
        self.tokenizer = self.processor = self.hf_config = None
        self.train_dataloader, self.eval_dataloader = train_dataloader, None
        self.global_step = 0
        self.infer_batch = lambda batch: {k: v for k, v in batch.items() if k != 'meta_info'}
        self.init_check()
        self.init_models(state)
        self.init_datasets()
        self.init_engines()
        self.init_logger()
        from .common import resume_from_slice
        resume_from_slice(self)

    # ------------------------------------------------------------------ init_* (rm.py:57-95)
    def init_check(self) -> None:
        from .common import refuse_unsupported_options
        refuse_unsupported_options(self.cfgs)
        if self.model_cfg is None and not cfg_get(self.cfgs, 'model_cfgs.model_name_or_path', None):
            raise ValueError('RMTrainer: model_cfg or model_cfgs.model_name_or_path is required')
        self.regularization = float(cfg_get(self.cfgs, 'train_cfgs.regularization', 0.001))

    def init_models(self, state=None) -> None:
        dt = compute_dtype(cfg_get(self.cfgs, 'train_cfgs.compute_dtype', 'bf16'))
        if self.model_cfg is None:
            from ..checkpoint import load_pretrained
            self.module, self.tokenizer, self.processor, self.hf_config = load_pretrained(
                cfg_get(self.cfgs, 'model_cfgs.model_name_or_path', None), self.device, trainable=True, head='score', dtype=dt,
                model_max_length=int(cfg_get(self.cfgs, 'model_cfgs.model_max_length', 512)), padding_side='right')
            self.model_cfg = self.module.cfg
            self.pad_token_id = getattr(self.tokenizer, 'pad_token_id', None)
            self._from_path = True
            return
        self._from_path = False
        self.module = build_model(self.model_cfg, self.device, trainable=True, head='score', dtype=dt)
        if state is not None:
            self.module.load_state_dict(state)

    def init_datasets(self) -> None:
        """rm.py:87-91 `get_dataloaders(PreferenceDataset, PreferenceDataset)` through the reference's own dataset / template plugins
        (common.get_dataloaders) when the model came from a directory (its tokenizer is needed) and no dataloader was handed in."""
        if self.train_dataloader is None and self._from_path:
            from .common import get_dataloaders
            self.train_dataloader, self.eval_dataloader = get_dataloaders(self, *self.dataset_types)

    def init_engines(self) -> None:
        # base/supervised_trainer.py:236-257: epochs x ceil(len(dataloader) / gas) updates -- known once train() has the dataloader
        t = lambda k, d: cfg_get(self.cfgs, 'train_cfgs.' + k, d)
        self.gas = int(t('gradient_accumulation_steps', cfg_get(self.ds_train_cfgs, 'gradient_accumulation_steps', 1)))
        total = t('total_training_steps', None)
        total = None if total is None else int(total)
        self.model = NativeEngine(self.module, lr=float(t('learning_rate', 2e-5)), betas=[float(b) for b in t('adam_betas', [0.9, 0.95])],
                                  weight_decay=float(t('weight_decay', 0.1)), max_grad_norm=float(cfg_get(self.ds_train_cfgs, 'gradient_clipping', 1.0)),
                                  total_steps=total, warmup_steps=int(float(t('lr_warmup_ratio', 0.03)) * (total or 0)),
                                  lr_scheduler_type=t('lr_scheduler_type', 'cosine'), gradient_accumulation_steps=self.gas)

    def init_logger(self) -> None:
        self.logger = None          # observability is out of scope; train() returns the metrics

    def _end_window(self, input_ids, attention_mask):
        """One row per sequence: the backbone's end position (device-side index math, no host sync)."""
        N, T = input_ids.shape
        dev = input_ids.device
        end = end_index(self.model.module.kind, attention_mask)
        rows_pad = pad64(N)
        row_idx = torch.zeros(rows_pad, dtype=torch.int64, device=dev)
        row_idx[:N] = torch.arange(N, device=dev) * T + end
        Mp = (N * T + 63) // 64 * 64
        inv = torch.full((Mp,), -1, dtype=torch.int32, device=dev)
        inv[row_idx[:N]] = torch.arange(N, dtype=torch.int32, device=dev)
        return {'N': N, 'T': T, 'rows': N, 'rows_pad': rows_pad, 'row_idx': row_idx, 'inv_map': inv,
                'labels': torch.zeros(rows_pad, dtype=torch.int64, device=dev)}, end

    def _mm(self, batch):
        """The multimodal keys of a batch for `response_scores`, the same for loss() and eval().  The vision-language reward models read
        the end score at position -1 even when that position is padding (models/llava.py:64-68, qwen2_vl.py:61-64), so what a masked
        query row sees matters: HF hides the right-padded keys from it -> kv_len = one past the last attended key, per row."""
        ids, am = batch['input_ids'], batch['attention_mask']
        mm = {k: batch[k] for k in ('image_grid_thw', 'position_ids3', 'input_features', 'feature_attention_mask') if k in batch}
        if self.model.module.kind in END_AT_LAST_POSITION_KINDS:
            T = ids.shape[1]
            mm['kv_len'] = (am.to(torch.int32) * torch.arange(1, T + 1, dtype=torch.int32, device=ids.device)).amax(dim=1).to(torch.int32)
        return mm

    def loss(self, batch):
        ids, am = batch['input_ids'], batch['attention_mask']
        B = ids.shape[0] // 2
        w, end = self._end_window(ids, am)
        self.model.wait_optimizer()
        mm = self._mm(batch)
        end_scores, scores = self.model.module.response_scores(ids, am, w, pixel_values=batch.get('pixel_values'), save=True,
                                                               all_scores=True, **mm)
        out2, d = ops.rm_loss(end_scores[:2 * B].contiguous(), B, self.regularization)
        dpad = torch.zeros(w['rows_pad'], dtype=torch.float32, device=ids.device)
        dpad[:2 * B] = d
        self.model.set_pending(dpad)
        # the six outputs of rm.py:125-132 (+ the fused [loss, accuracy] pair the step all-reduces once)
        return {'loss': out2[0], 'accuracy': out2[1], 'higher_end_reward': end_scores[:B], 'lower_end_reward': end_scores[B:2 * B],
                'higher_rewards': scores[:B], 'lower_rewards': scores[B:2 * B], '_stats': out2}

    def train_step(self, batch):
        ld = self.loss(batch)
        self.model.backward(ld['loss'])
        self.model.step()
        s = get_all_reduce_mean(ld['_stats'].clone()).tolist()
        return {'train/loss': s[0], 'train/accuracy': s[1], 'train/lr': self.model.optimizer.param_groups[0]['lr']}

    def train(self, train_dataloader=None) -> list:
        """rm.py:265-330 without its logging: `epochs` passes of `train_step`; resumes at `self.global_step` (remaining epochs, the first
        `global_step % len(dataloader)` batches of the resumed epoch skipped, :276-292) and saves `slice_<global_step>` every
        epochs * len(dataloader) // logger_cfgs.save_total_limit steps (:307-314); with `data_cfgs.eval_datasets` it evaluates before the first step, every
        `eval_interval` steps under eval_strategy 'steps' and after every epoch (:274-325; results in `self.eval_history` as (global_step, dict)).
        Returns the per-step metrics."""
        dl = train_dataloader if train_dataloader is not None else getattr(self, 'train_dataloader', None)
        if dl is None:
            raise ValueError('RMTrainer.train needs a dataloader of preference batches')
        history = []
        self.global_step = getattr(self, 'global_step', 0)
        epochs = int(cfg_get(self.cfgs, 'train_cfgs.epochs', 1))
        n = len(dl) if hasattr(dl, '__len__') else None
        if self.model.total_steps is None and n is not None:
            self.model.set_schedule(epochs * ((n + self.gas - 1) // self.gas), float(cfg_get(self.cfgs, 'train_cfgs.lr_warmup_ratio', 0.03)))
        remain = epochs - self.global_step // n if n else epochs
        skip = self.global_step % n if n else 0
        every = save_interval(self.cfgs, epochs * n if n else None)
        self.eval_history = getattr(self, 'eval_history', [])
        if eval_due(self.cfgs, 'begin'):                                  # rm.py:274-275
            self.eval_history.append((0, self.eval()))
        for epoch in range(int(remain)):
            for i, batch in enumerate(dl):
                if epoch == 0 and i < skip:
                    continue
                history.append(self.train_step(batch))
                self.global_step += 1
                if every and self.global_step % every == 0:
                    self.save(tag=self.global_step)
                if eval_due(self.cfgs, 'steps', self.global_step):        # rm.py:316-322
                    self.eval_history.append((self.global_step, self.eval()))
            if eval_due(self.cfgs, 'always'):                             # rm.py:324-325: after every epoch
                self.eval_history.append((self.global_step, self.eval()))
        return history

    def save(self, model=None, tag=None, output_dir=None) -> str:
        """base/supervised_trainer.py:404-450 (common.save_slice): the score model in the layout `from_pretrained` loads."""
        return save_slice(self, model or self.model, tag, output_dir)

    @torch.no_grad()
    def eval(self, eval_dataloader=None) -> dict:
        """rm.py eval (the loop before `train_step` in the reference file): end scores of every chosen / rejected pair of the evaluation
        set, `eval/accuracy` (chosen scored higher; all-reduce mean like the reference), `eval/reward_mean` / `eval/reward_std` over the rewards
        of all ranks (the reference gathers them on rank 0; here every rank gets the same numbers).  Returns {} without a dataloader."""
        import torch.distributed as dist
        dl = eval_dataloader if eval_dataloader is not None else getattr(self, 'eval_dataloader', None)
        if dl is None:
            return {}
        correct, total, rewards = None, 0, []
        for batch in dl:
            ids, am = batch['input_ids'], batch['attention_mask']
            B = ids.shape[0] // 2
            w, _ = self._end_window(ids, am)
            self.model.wait_optimizer()
            mm = self._mm(batch)
            s = self.model.module.response_scores(ids, am, w, pixel_values=batch.get('pixel_values'), save=False, **mm)[:2 * B]
            hits = (s[:B] > s[B:]).sum()
            correct = hits if correct is None else correct + hits
            total += B
            rewards.append(s.float())
        if not rewards:
            return {}
        accuracy = get_all_reduce_mean((correct.float() / total).reshape(1))
        rewards = torch.cat(rewards)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            parts = [torch.empty_like(rewards) for _ in range(dist.get_world_size())]
            dist.all_gather(parts, rewards)
            rewards = torch.cat(parts)
        return {'eval/accuracy': float(accuracy.item()), 'eval/reward_mean': float(rewards.mean().item()),
                'eval/reward_std': float(rewards.std().item())}

