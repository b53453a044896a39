"""Native DPO trainer -- the reference's DPOTrainer surface over the MI355X hot path.

Mirrors align_anything/trainers/text_to_text/dpo.py:57-321 and trainers/text_image_to_text/dpo.py:50-166:
same method set (init_models / init_engines / compute_log_probs / loss / train_step / train / eval / save),
same batch contract (PreferenceCollator: rows [0,B) chosen, [B,2B) rejected, left padded, meta_info.response_lens),
same metric keys.  What differs is what executes: policy/reference forwards, the log-prob gather, the DPO
loss (+ its gradient), backward, clip, AdamW and the gradient all-reduce are HIP kernels / RCCL behind
include/aa_hip.h; python only sequences them.

Deliberate numerical difference: the reference sums bf16 log-probs in bf16, which quantises
sum(logp) to 8 mantissa bits; here per-token log-probs are fp32 and reduced in fp32.  `emulate_bf16_logp=True`
reproduces the reference's bf16 rounding of the per-token values for A/B checks.
"""
from __future__ import annotations

import os
from typing import Any

import torch

from .. import ops
from ..engine import NativeEngine
from ..modeling import build_model
from .common import build_pack_plan, build_window, cfg_get, compute_dtype, expert_parallel_kwargs, flat_to_padded, get_all_reduce_mean, save_slice


class DPOTrainer:
    uses_reference = True     # SimPO / ORPO (trainers/pref.py) never evaluate the reference model
    skip_identical_pairs = False   # set by __init__ for the audio trainer, whose reference `loss` skips identical pairs
    dataset_types = ('PreferenceDataset', 'PreferenceDataset')      # (train, eval) classes of align_anything.datasets.<modality> (dpo.py:109-113)

    def __init__(self, cfgs, ds_cfgs=None, *, model_cfg: dict | None = None, policy_state=None, reference_state=None,
                 train_dataloader=None, tokenizer=None, device='cuda:0', share_vision_tower=True,
                 emulate_bf16_logp=False, dtype=None):
        self.cfgs, self.ds_train_cfgs = cfgs, ds_cfgs
        # compute dtype: bf16 (the reference's `bf16: True`, configs/train/*/dpo.yaml) or fp32 = parity mode that
        # tracks the reference's fp32 CPU trainer (include/aa_hip_f32.h); also selectable as train_cfgs.compute_dtype
        self.dtype = compute_dtype(dtype if dtype is not None else cfg_get(cfgs, 'train_cfgs.compute_dtype', 'bf16'))
        self.device = torch.device(device)
        self.model_cfg = model_cfg
        self.train_dataloader, self.eval_dataloader = train_dataloader, None
        self.tokenizer, self.processor, self.hf_config = tokenizer, None, None
        self.global_step = 0
        self.emulate_bf16_logp = emulate_bf16_logp
        self.share_vision_tower = share_vision_tower
        # shared-prompt packing (trainers.common.build_pack_plan): the pair's common prefix is computed once per model instead of once per row.  Opt-in
        # (train_cfgs.share_prompt_prefix / AA_SHARE_PROMPT=1): same results up to the stated rounding, 62.5 % of the token rows at T = 2048, R = 512
        self.share_prompt_prefix = bool(cfg_get(cfgs, 'train_cfgs.share_prompt_prefix', os.environ.get('AA_SHARE_PROMPT', '0') == '1'))
        # trainers/text_audio_to_text/dpo.py:139-140 `continue`s on identical pairs; the text and text+image trainers
        # (text_to_text/dpo.py:160-176, text_image_to_text/dpo.py:134-153) do not
        self.skip_identical_pairs = bool(model_cfg and model_cfg.get('kind') == 'qwen2audio')
        self.infer_batch = lambda batch: {k: v for k, v in batch.items() if k != 'meta_info'}
        # the reference's constructor order (text_to_text/dpo.py:59-77): init_check, init_models, init_datasets, init_engines, init_logger.
        # `DPOTrainer(cfgs, ds_cfgs)` alone works like the reference's: the models come from model_cfgs.model_name_or_path, the
        # dataloaders from data_cfgs; the keyword arguments inject pre-built pieces instead (tests, bench.py, INTEGRATION.md level B)
        # everything that can refuse a configuration runs BEFORE both models are streamed into HBM; only what needs the loaded
        # tokenizer / config (pad_token_id) waits for init_models
        from .common import refuse_unsupported_options
        refuse_unsupported_options(self.cfgs)
        if self.model_cfg is None and not cfg_get(self.cfgs, 'model_cfgs.model_name_or_path', None):
            raise ValueError('model_cfg (align_anything_amd.configs dict) or model_cfgs.model_name_or_path is required')
        self.init_models(policy_state, reference_state)
        self.init_check()
        self.init_datasets()
        self.init_engines()
        self.init_logger()
        from .common import resume_from_slice
        resume_from_slice(self)

    # ------------------------------------------------------------------ init_*
    def init_check(self) -> None:
        from .common import refuse_unsupported_options
        refuse_unsupported_options(self.cfgs)
        if self.model_cfg is None:
            raise ValueError('model_cfg (align_anything_amd.configs dict) or model_cfgs.model_name_or_path is required')
        self.scale_coeff = float(cfg_get(self.cfgs, 'train_cfgs.scale_coeff', 0.1))
        self.pad_token_id = cfg_get(self.cfgs, 'model_cfgs.pad_token_id', None)
        if self.pad_token_id is None:
            self.pad_token_id = getattr(self.tokenizer, 'pad_token_id', None)
        if self.pad_token_id is None:
            self.pad_token_id = self.model_cfg.get('pad_token_id')
        if self.pad_token_id is None:
            raise ValueError('pad_token_id is required (tokenizer.pad_token_id or model_cfgs.pad_token_id)')

    def init_models(self, policy_state=None, reference_state=None) -> None:
        """text_image_to_text/dpo.py:58-83: policy with freeze flags, frozen reference from the same checkpoint.  Without a `model_cfg` the
        geometry, the weights of both models, the tokenizer and the processor come from `model_cfgs.model_name_or_path`
        (checkpoint.load_pretrained = the reference's load_pretrained_models, models/pretrained_model.py:160-312)."""
        from_path = self.model_cfg is None
        if from_path:
            from transformers import AutoConfig
            from .. import configs
            path = cfg_get(self.cfgs, 'model_cfgs.model_name_or_path', None)
            if not path:
                raise ValueError('model_cfg (align_anything_amd.configs dict) or model_cfgs.model_name_or_path is required')
            self.model_cfg = configs.from_hf_config(AutoConfig.from_pretrained(path, trust_remote_code=True))     # kind -> freeze flags below
            self.skip_identical_pairs = self.model_cfg.get('kind') == 'qwen2audio'
        freeze = {}
        if self.model_cfg['kind'] == 'qwen2audio':
            freeze = dict(freeze_mm_proj=bool(cfg_get(self.cfgs, 'train_cfgs.freeze_mm_proj', False)),
                          freeze_language_model=bool(cfg_get(self.cfgs, 'train_cfgs.freeze_language_model', False)),
                          freeze_audio_tower=bool(cfg_get(self.cfgs, 'train_cfgs.freeze_audio_tower', False)))
        if self.model_cfg['kind'] in ('llava', 'qwen2vl'):
            freeze = dict(freeze_mm_proj=bool(cfg_get(self.cfgs, 'train_cfgs.freeze_mm_proj', False)),
                          freeze_language_model=bool(cfg_get(self.cfgs, 'train_cfgs.freeze_language_model', False)),
                          freeze_vision_tower=bool(cfg_get(self.cfgs, 'train_cfgs.freeze_vision_tower', True)))
        ref_kw = {}
        epk = expert_parallel_kwargs(self.cfgs, self.model_cfg)
        if epk:
            # experts split over the data-parallel ranks (expert_parallel.py); the exchange runs on its own communicator so it
            # never queues behind a gradient bucket.  Policy and reference shard the same way.
            freeze = ref_kw = epk
        if from_path:
            from .common import resolve_pretrained
            self.policy, tok, self.processor, self.hf_config = resolve_pretrained(self.cfgs, self.device, trainable=True, dtype=self.dtype, build_kwargs=freeze)
            self.tokenizer = self.tokenizer or tok
            self.model_cfg = self.policy.cfg          # after the pad-token resize (vocab + 1 when the tokenizer had no pad token)
            self.reference = (resolve_pretrained(self.cfgs, self.device, trainable=False, dtype=self.dtype, build_kwargs=ref_kw)[0]
                              if self.uses_reference else None)
            return
        self.policy = build_model(self.model_cfg, self.device, trainable=True, dtype=self.dtype, **freeze)
        self.reference = build_model(self.model_cfg, self.device, trainable=False, dtype=self.dtype, **ref_kw) if self.uses_reference else None
        if policy_state is not None:
            self.policy.load_state_dict(policy_state)
            if self.reference is not None:
                self.reference.load_state_dict(reference_state if reference_state is not None else policy_state)

    def init_datasets(self) -> None:
        """text_to_text/dpo.py:109-113 `get_dataloaders(PreferenceDataset, PreferenceDataset)`: built from data_cfgs through the reference's own
        dataset / template plugins (common.get_dataloaders) unless a dataloader was handed in or data_cfgs names no dataset."""
        if self.train_dataloader is not None:
            return
        from .common import get_dataloaders
        self.train_dataloader, self.eval_dataloader = get_dataloaders(self, *self.dataset_types)

    def init_engines(self) -> None:
        """base/supervised_trainer.py:234-271 + dpo.py:114-120, with the native engine in DeepSpeed's place."""
        t = lambda k, d: cfg_get(self.cfgs, 'train_cfgs.' + k, d)
        gas = int(t('gradient_accumulation_steps', 1))
        total = None         # unknown: a cosine schedule refuses to step (engine._lr_at) instead of decaying to 0 after one update
        if self.train_dataloader is not None and hasattr(self.train_dataloader, '__len__'):
            total = int(t('epochs', 1)) * ((len(self.train_dataloader) + gas - 1) // gas)      # supervised_trainer.py:236-239
        total = t('total_training_steps', total)
        total = int(total) if total is not None else None
        betas = [float(b) for b in t('adam_betas', [0.9, 0.95])]
        self.model = NativeEngine(self.policy, lr=float(t('learning_rate', 1e-6)), betas=betas,
                                  eps=float(t('adam_epsilon', 1e-8)), weight_decay=float(t('weight_decay', 0.0)),
                                  max_grad_norm=float(cfg_get(self.ds_train_cfgs, 'gradient_clipping', 1.0)),
                                  total_steps=total, warmup_steps=int(float(t('lr_warmup_ratio', 0.03)) * (total or 0)),
                                  lr_scheduler_type=t('lr_scheduler_type', 'cosine'), trainable=True,
                                  gradient_accumulation_steps=gas)
        self.reference_model = NativeEngine(self.reference, trainable=False) if self.reference is not None else None

    def init_logger(self) -> None:
        self.logger = None  # observability is out of scope; train() returns the metrics

    # ------------------------------------------------------------------ hot path
    def _features(self, batch):
        """Frozen CLIP tower once per unique image (collator stacks images*2: rows [0,B) == rows [B,2B),
        datasets/text_image_to_text/preference.py:219-222), shared by policy and reference."""
        pv = batch.get('pixel_values')
        if pv is None or self.policy.kind != 'llava' or getattr(self.policy, 'train_tower', False):
            return None      # a training tower is run by the policy itself (its activations are needed) and differs from the reference's
        if '_vision_features' in batch:
            return batch['_vision_features']
        tower = self.policy if self.share_vision_tower else None
        n = pv.shape[0]
        if tower is not None and n % 2 == 0:
            f = tower.vision_features(pv[: n // 2])
            batch['_vision_features_unique'] = f          # one set per pair: what the packed layout consumes (its image tokens appear once per pair)
            f = torch.cat([f, f], 0)
        else:
            f = None
        batch['_vision_features'] = f
        return f

    def _pack_plan(self, batch):
        """train_cfgs.share_prompt_prefix (default off): trainers.common.build_pack_plan for this batch, or None when the model / batch does not qualify
        (LLaVA, Llama-family, Qwen2-VL, Qwen2-Audio and single-rank Qwen3-MoE decoders in the left-padded pair layout; a training vision tower or unshared images / clips keep the reference layout)."""
        if not self.share_prompt_prefix:
            return None
        if '_pack' not in batch:
            plan = None
            kind = self.policy.kind
            ok = kind in ('llava', 'llama', 'qwen2audio', 'qwen2vl', 'qwen3moe') and not getattr(self.policy, 'tied', False) \
                and not (kind == 'llava' and getattr(self.policy, 'train_tower', False)) and not (kind == 'qwen3moe' and getattr(self.policy, 'ep', None) is not None)
            fa = batch.get('input_features')
            if ok and kind == 'qwen2audio' and fa is not None:       # the clips are stacked twice like the images (one device read unless the collator vouches)
                fm, h = batch.get('feature_attention_mask'), fa.shape[0] // 2
                ok = fa.shape[0] % 2 == 0 and ('shared_prefix_lens' in batch['meta_info']
                                               or (bool(torch.equal(fa[:h], fa[h:])) and (fm is None or bool(torch.equal(fm[:h], fm[h:])))))
            pv = batch.get('pixel_values')
            if ok and pv is not None and kind == 'qwen2vl':           # flattened patches of all images + their grids: both stacked twice (the module runs its own tower, on one half)
                grid = batch.get('image_grid_thw')
                grid = grid.tolist() if isinstance(grid, torch.Tensor) else [list(g) for g in (grid or [])]
                ok = len(grid) > 0 and len(grid) % 2 == 0 and grid[:len(grid) // 2] == grid[len(grid) // 2:] and pv.shape[0] % 2 == 0 \
                    and ('shared_prefix_lens' in batch['meta_info'] or bool(torch.equal(pv[:pv.shape[0] // 2], pv[pv.shape[0] // 2:])))
            elif ok and pv is not None:
                ok = self.share_vision_tower and self._features(batch) is not None
                # image placeholder ids look alike whatever the image: rows may only share their image positions when they carry the SAME image.  A collator that
                # states the shared prefix (meta_info.shared_prefix_lens) vouches for it (the reference's stacks `images * 2`); otherwise look (one device read)
                if ok and 'shared_prefix_lens' not in batch['meta_info']:
                    ok = pv.shape[0] % 2 == 0 and bool(torch.equal(pv[:pv.shape[0] // 2], pv[pv.shape[0] // 2:]))
            if ok:
                plan = build_pack_plan(batch['input_ids'], batch.get('attention_mask'), self._window(batch), batch['meta_info'])
                # The multimodal forward feeds ONE set of features per pair, so every image / audio placeholder of a pair must lie inside its shared prefix
                # (a pair without a common prefix, or whose rows part before the placeholders, would need its own copy): count them in the packed ids -- one
                # device read, skipped when the collator states the shared prefix.
                tok = self.policy.cfg.get('image_token_id', self.policy.cfg.get('audio_token_id')) if isinstance(getattr(self.policy, 'cfg', None), dict) else None
                if plan is not None and tok is not None and (pv is not None or fa is not None):
                    if min(plan['prefix_lens']) == 0:
                        plan = None
                    elif 'shared_prefix_lens' not in batch['meta_info']:
                        ids = batch['input_ids']
                        if int((plan['ids'] == tok).sum()) != int((ids[:ids.shape[0] // 2] == tok).sum()):
                            plan = None
            batch['_pack'] = plan
        return batch['_pack']

    def _window(self, batch):
        if '_window' not in batch:
            batch['_window'] = build_window(batch['input_ids'], batch['meta_info']['response_lens'], self.pad_token_id)
        return batch['_window']

    def _flat_log_probs(self, module, batch, save):
        w = self._window(batch)
        feats = self._features(batch) if (self.share_vision_tower or module is self.policy) else None
        pack = self._pack_plan(batch)
        if pack is not None:
            mm = {k: batch[k] for k in ('image_grid_thw', 'position_ids3', 'input_features', 'feature_attention_mask') if k in batch}
            return module.response_logprobs(batch['input_ids'], batch.get('attention_mask'), w, save=save, round_bf16=self.emulate_bf16_logp, pack=pack,
                                            pixel_values=batch.get('pixel_values') if module.kind == 'qwen2vl' else None,
                                            image_features=batch.get('_vision_features_unique') if feats is not None else None, **mm)
        mm = {k: batch[k] for k in ('image_grid_thw', 'position_ids3', 'input_features', 'feature_attention_mask') if k in batch}   # Qwen2-VL / Qwen2-Audio processor outputs
        return module.response_logprobs(batch['input_ids'], batch.get('attention_mask'), w,
                                        pixel_values=batch.get('pixel_values') if feats is None else None,
                                        save=save, image_features=feats, round_bf16=self.emulate_bf16_logp, **mm)

    def compute_log_probs(self, model, batch) -> torch.Tensor:
        """dpo.py:122-142: [2B, max(R)-1] response-window log-probs, right-padded with 0.0 (fp32 here)."""
        module = getattr(model, 'module', model)
        if hasattr(model, 'wait_optimizer'):
            model.wait_optimizer()
        flat = self._flat_log_probs(module, batch, save=False)
        return flat_to_padded(flat, self._window(batch))

    def loss(self, batch) -> dict[str, torch.Tensor]:
        """dpo.py:144-203.  Also stages d loss / d logp for engine.backward (the loss kernel emits both)."""
        w = self._window(batch)
        B = w['N'] // 2
        # reference first: it does not read the policy, so it overlaps with the previous step's asynchronous
        # optimizer update (NativeEngine.step); the policy forward then joins on it
        ref = self._flat_log_probs(self.reference_model.module, batch, save=False)
        self.model.wait_optimizer()
        pol = self._flat_log_probs(self.model.module, batch, save=True)
        keep = None
        if self.skip_identical_pairs:   # text_audio_to_text/dpo.py:139-140: identical pairs contribute nothing
            am = batch.get('attention_mask')
            _, _, _, keep = ops.pair_slice_index(batch['input_ids'], am if am is not None else torch.ones_like(batch['input_ids']), w['seq_off'], B)
        out6, per, dlogp = ops.dpo_loss(pol, ref, w['seq_off'], B, self.scale_coeff, want_grad=True, keep=keep)
        self.model.set_pending(dlogp)
        sel = (lambda t: t[keep.bool()]) if keep is not None else (lambda t: t)
        return {
            'loss': out6[0], 'reward': sel(per[2]), 'better_sample_reward': sel(per[0]), 'worse_sample_reward': sel(per[1]),
            'reward_accuracy': out6[1], 'reward_margin': sel(per[3]), '_means': out6,
        }

    def train_step(self, batch) -> dict[str, Any]:
        """dpo.py:205-237; the six scalar all-reduces + six .item() syncs are fused into one each."""
        loss_dict = self.loss(batch)
        if not getattr(self, '_first_batch_checked', False):
            self._first_batch_checked = True
            self._check_first_batch(batch)
        self.model.backward(loss_dict['loss'])
        self.model.step()
        means = get_all_reduce_mean(loss_dict['_means'].clone())
        m = means.tolist()  # the single device->host sync of the step
        return {
            'train/loss': m[0], 'train/reward': m[2], 'train/better_sample_reward': m[3],
            'train/worse_sample_reward': m[4], 'train/reward_accuracy': m[1], 'train/reward_margin': m[5],
            'train/lr': self.model.optimizer.param_groups[0]['lr'],
        }

    def _check_first_batch(self, batch) -> None:
        """Once per trainer, after the first forward (two host syncs that the hot loop must not pay per step; AA_VALIDATE_FIRST_BATCH=0 skips them):
        (1) what hf raises on every forward -- "Image features and image tokens do not match" (hf:models/llava/modeling_llava.py:191-213; a
        processor that did not expand `<image>`, a truncated prompt) -- instead of scattering a wrong number of feature rows silently;
        (2) the shared-tower shortcut of `_features` rests on the reference collator's layout (`images * 2`: the rejected row carries the chosen
        row's image, datasets/text_image_to_text/preference.py:219-222): a batch whose halves differ needs `share_vision_tower=False`."""
        import os
        if os.environ.get('AA_VALIDATE_FIRST_BATCH', '1') == '0':
            return
        mod = self.model.module
        if getattr(mod, '_last_image_token_count', None) is not None and hasattr(mod, 'validate_batch'):
            mod.validate_batch()
        pv = batch.get('pixel_values')
        if (pv is not None and self.policy.kind == 'llava' and self.share_vision_tower and not getattr(self.policy, 'train_tower', False)
                and pv.shape[0] % 2 == 0 and not torch.equal(pv[:pv.shape[0] // 2], pv[pv.shape[0] // 2:])):
            raise ValueError('DPOTrainer: the images of the chosen and the rejected rows differ, but the vision tower is run once per pair (the reference collator '
                             'stacks `images * 2`); construct the trainer with share_vision_tower=False for batches laid out differently')

    def train(self) -> list[dict[str, Any]]:
        """dpo.py:239-308 without the per-step torch_gc() (a ZeRO-3 memory work-around): resumes at
        `self.global_step` (remaining epochs, the first `global_step % len(dataloader)` batches of the first one skipped,
        dpo.py:256-270), saves `slice_<global_step>` every epochs * len(dataloader) // logger_cfgs.save_total_limit steps
        (dpo.py:285-293) and evaluates on the reference's `eval_strategy` schedule when `data_cfgs.eval_datasets` is set."""
        history = []
        epochs = int(cfg_get(self.cfgs, 'train_cfgs.epochs', 1))
        n_batches = len(self.train_dataloader) if hasattr(self.train_dataloader, '__len__') else None
        if self.model.total_steps is None and n_batches is not None:   # dataloader attached after __init__
            self.model.set_schedule(epochs * ((n_batches + self.model.gas - 1) // self.model.gas),
                                    float(cfg_get(self.cfgs, 'train_cfgs.lr_warmup_ratio', 0.03)))
        per_epoch = max(1, n_batches or 1)
        remain_epoch = epochs - self.global_step // per_epoch if n_batches is not None else epochs
        start_batch_idx = self.global_step % per_epoch if n_batches is not None else 0
        limit = cfg_get(self.cfgs, 'logger_cfgs.save_total_limit', None)
        save_interval = (epochs * per_epoch // int(limit)) if (limit and n_batches is not None) else 0
        evals = bool(cfg_get(self.cfgs, 'data_cfgs.eval_datasets', None))
        strategy = cfg_get(self.cfgs, 'train_cfgs.eval_strategy', 'epoch')
        eval_interval = int(cfg_get(self.cfgs, 'train_cfgs.eval_interval', 0) or 0)
        if evals:
            history.append({'eval/step': 0, **self.eval()})
        for epoch in range(int(remain_epoch)):
            self.model.train()
            self._epoch_begin()
            for batch_idx, batch in enumerate(self.train_dataloader):
                if epoch == 0 and batch_idx < start_batch_idx:
                    continue
                info = self.train_step(batch)
                self.global_step += 1
                info['train/epoch'] = self.global_step / per_epoch
                history.append(info)
                if save_interval > 0 and self.global_step % save_interval == 0:
                    self.save(tag=self.global_step)
                if evals and strategy == 'steps' and eval_interval > 0 and self.global_step % eval_interval == 0:
                    history.append({'eval/step': self.global_step, **self.eval()})
            if evals and strategy == 'epoch':
                history.append({'eval/step': self.global_step, **self.eval()})
            self.model.tput_timer.update_epoch_count()
        return history

    def _epoch_begin(self) -> None:
        """Hook at the start of every epoch of `train()` (KTO refreshes its KL estimate here, kto.py:211-215)."""

    def eval(self) -> dict[str, Any]:
        return {}  # the reference's DPO eval is a stub (dpo.py:310-313)

    def save(self, model=None, tag=None, output_dir=None) -> str:
        """base/supervised_trainer.py:404-450: <output_dir>/slice_<tag|end>/ in the layout `from_pretrained` loads (common.save_slice)."""
        return save_slice(self, model or self.model, tag, output_dir)
