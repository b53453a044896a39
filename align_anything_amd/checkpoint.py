"""Reader for HuggingFace checkpoints: `model_cfgs.model_name_or_path` -> the flat device buffers of the native model.

The reference loads its models itself: `load_pretrained_models(cfgs.model_cfgs.model_name_or_path, ...)`
(align_anything/models/pretrained_model.py:160-312, called from trainers/text_to_text/dpo.py:83-100 and
trainers/text_image_to_text/dpo.py:58-83) = `AnyModel.from_pretrained` (bf16) + tokenizer + processor + pad-token resize
(pretrained_model.py:61-157).  This module is the native counterpart:

  * `LazyCheckpoint(path)`: a read-only mapping parameter name -> tensor over `model.safetensors`,
    `model-0000x-of-0000y.safetensors` + `model.safetensors.index.json`, `pytorch_model.bin` and its sharded form
    (`pytorch_model.bin.index.json`), resolved in `from_pretrained`'s order.  Nothing is read until a name is asked for;
    safetensors shards are memory-mapped and ONE tensor at a time is materialised, so `ParamStore.load_state_dict(lazy)` streams
    every tensor straight into its slice of the flat bf16 / fp32 buffers in HBM -- the host never holds the model (a 7B
    checkpoint is 13.5 GB; two replicas of it, policy + reference, load without 27 GB of host copies).
  * key layouts: checkpoints on the hub were written by transformers 4.x (`language_model.model.layers...`,
    `vision_tower.vision_model...`, `visual...`), the native names are the ones transformers >= 5 uses
    (`model.language_model.layers...`, `model.vision_tower...`, `model.visual...`); the renames below restate
    hf:conversion_mapping.py ("llava", "qwen2_audio", "Qwen2VLForConditionalGeneration", "CLIPVisionModel") so both load.
    Per-expert Qwen3-MoE keys are merged by the model (`NativeQwen3Moe.fuse_expert_keys`, lazily).
  * `load_pretrained(...)`: config.json -> native geometry, tokenizer / processor when their files exist, the pad-token
    resize of pretrained_model.py:112-150 (a new `<pad>` row = mean of the old rows, input and output embeddings),
    weights streamed into a freshly built native model.
"""
from __future__ import annotations

import json
import os
import re
from collections.abc import Mapping

import torch

DEFAULT_PAD_TOKEN = '<pad>'          # align_anything/utils/template_registry / pretrained_model.py:45 DEFAULT_PAD_TOKEN

# (pattern, replacement) applied in order to every checkpoint key; the result is the transformers >= 5 name the native stores use
_RENAMES = {
    'llava': [(r'^language_model\.lm_head', 'lm_head'), (r'^language_model\.model', 'model.language_model'),
              (r'^vision_tower', 'model.vision_tower'), (r'^multi_modal_projector', 'model.multi_modal_projector'),
              (r'^model\.vision_tower\.vision_model\.', 'model.vision_tower.')],
    # (transformers 5.x `save_pretrained` reverses TWO of its renames for this model and writes `language_model.model.model.<...>`: accepted as well)
    'qwen2audio': [(r'^language_model\.lm_head', 'lm_head'), (r'^language_model\.model\.model\.', 'model.language_model.'), (r'^language_model\.model', 'model.language_model'),
                   (r'^audio_tower', 'model.audio_tower'), (r'^multi_modal_projector', 'model.multi_modal_projector')],
    'qwen2vl': [(r'^visual', 'model.visual'), (r'^model(?!\.(language_model|visual))', 'model.language_model')],
    'opt': [(r'^decoder\.', 'model.decoder.')],
    'llama': [],
    'qwen3moe': [],
}


def normalize_key(kind: str, key: str) -> str:
    for pat, rep in _RENAMES.get(kind, []):
        key = re.sub(pat, rep, key, count=1)
    return key


def _resolve_files(path: str):
    """(format, {file: [keys] or None}) in from_pretrained's preference order (hf:modeling_utils.py _get_resolved_checkpoint_files):
    model.safetensors, model.safetensors.index.json, pytorch_model.bin, pytorch_model.bin.index.json."""
    p = lambda n: os.path.join(path, n)

    def sharded(index):
        with open(p(index)) as f:
            wm = json.load(f)['weight_map']
        files = {}
        for k, fn in wm.items():
            files.setdefault(p(fn), []).append(k)
        missing = [fn for fn in files if not os.path.isfile(fn)]
        if missing:
            raise FileNotFoundError(f'{index} names shards that do not exist: {missing[:3]}')
        return files

    if os.path.isfile(p('model.safetensors')):
        return 'safetensors', {p('model.safetensors'): None}
    if os.path.isfile(p('model.safetensors.index.json')):
        return 'safetensors', sharded('model.safetensors.index.json')
    if os.path.isfile(p('pytorch_model.bin')):
        return 'bin', {p('pytorch_model.bin'): None}
    if os.path.isfile(p('pytorch_model.bin.index.json')):
        return 'bin', sharded('pytorch_model.bin.index.json')
    raise FileNotFoundError(f'no model.safetensors / model.safetensors.index.json / pytorch_model.bin(.index.json) under {path!r}')


class LazyCheckpoint(Mapping):
    """name (transformers >= 5 layout of `kind`) -> CPU tensor, read from disk when asked for."""

    def __init__(self, path: str, kind: str = 'llama'):
        self.path, self.kind = path, kind
        self.format, files = _resolve_files(path)
        self._where = {}                       # normalised name -> (file, key in the file)
        self._open = {}                        # file -> handle (safe_open) / dict (torch.load mmap); one .bin shard at a time
        for fn, keys in files.items():
            if keys is None:
                keys = self._keys_of(fn)
            for k in keys:
                self._where[normalize_key(kind, k)] = (fn, k)

    def _handle(self, fn):
        h = self._open.get(fn)
        if h is None:
            if self.format == 'safetensors':
                from safetensors import safe_open
                h = safe_open(fn, framework='pt', device='cpu')
            else:
                self._open.clear()             # a .bin shard is one pickle: keep a single one mapped
                h = torch.load(fn, map_location='cpu', mmap=True, weights_only=True)
            self._open[fn] = h
        return h

    def _keys_of(self, fn):
        return list(self._handle(fn).keys())

    def __getitem__(self, name):
        fn, key = self._where[name]
        h = self._handle(fn)
        return h.get_tensor(key) if self.format == 'safetensors' else h[key]

    def __iter__(self):
        return iter(self._where)

    def __len__(self):
        return len(self._where)

    def __contains__(self, name):
        return name in self._where

    def close(self):
        self._open.clear()


class Derived(Mapping):
    """A lazy mapping `base` with some names removed and some computed on access (thunks): how per-expert hub tensors become the fused
    [E, ...] blocks without the host ever holding more than one layer's experts (modeling.NativeQwen3Moe.fuse_expert_keys)."""

    def __init__(self, base, drop, thunks):
        self.base, self.thunks = base, dict(thunks)
        self._names = [k for k in base if not drop(k)] + list(self.thunks)
        self._set = set(self._names)

    def __getitem__(self, name):
        t = self.thunks.get(name)
        return t() if t is not None else self.base[name]

    def __iter__(self):
        return iter(self._names)

    def __len__(self):
        return len(self._names)

    def __contains__(self, name):
        return name in self._set


class _WithNewRows(Mapping):
    """The pad-token resize of pretrained_model.py:112-150 on a lazy checkpoint: `model.resize_token_embeddings(len(tokenizer))` followed by
    `init_new_embeddings` -- the named [vocab, ...] tensors are cut or grown to EXACTLY `rows` rows (a checkpoint whose embedding is larger than
    its tokenizer shrinks, as it does in the reference) and their last `extra` rows become the mean of the rows before them; everything else
    passes through."""

    def __init__(self, base, names, extra, rows=None):
        self.base, self.names, self.extra, self.rows = base, set(names), int(extra), rows

    def __getitem__(self, name):
        t = self.base[name]
        if name in self.names and self.extra > 0:
            rows = int(self.rows) if self.rows is not None else t.shape[0] + self.extra
            keep = t[:rows - self.extra] if rows - self.extra <= t.shape[0] else torch.cat(
                [t, torch.zeros((rows - self.extra - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype)], dim=0)
            mean = keep.float().mean(dim=0, keepdim=True).to(t.dtype)
            t = torch.cat([keep, mean.expand(self.extra, *t.shape[1:])], dim=0)
        return t

    def __iter__(self):
        return iter(self.base)

    def __len__(self):
        return len(self.base)

    def __contains__(self, name):
        return name in self.base


def _embedding_names(names):
    return [n for n in names if n.endswith('embed_tokens.weight') or n == 'lm_head.weight']


def check_vocab_rows(rows: int, path: str = '') -> None:
    """The native lm_head / log-prob kernels take vocabularies that are multiples of 4 (aa_lmhead_logprob_fwd).  The reference's pad-token resize
    can leave an odd row count (its own note: "may make your embedding size not be divisible by 64", pretrained_model.py:65): say so at load time,
    with the way out, instead of at the first step."""
    if rows % 4:
        raise RuntimeError(
            f'{path}: the tokenizer has no pad token; adding {DEFAULT_PAD_TOKEN!r} as the reference does (pretrained_model.py:112-150) resizes the embeddings to '
            f'{rows} rows, and the native lm_head / log-prob kernels take vocabularies that are multiples of 4.  Give the tokenizer a pad token that is already '
            'in its vocabulary (tokenizer_config.json "pad_token", e.g. the eos token): the embeddings then keep their size.')


def load_tokenizer_and_processor(path, model_max_length=512, padding_side='left', processor_kwargs=None):
    """pretrained_model.py:279-312: AutoTokenizer, then AutoProcessor (None when the directory has none); a processor's own tokenizer wins."""
    from transformers import AutoProcessor, AutoTokenizer
    tokenizer = processor = None
    try:
        tokenizer = AutoTokenizer.from_pretrained(path, model_max_length=model_max_length, padding_side=padding_side, trust_remote_code=True)
    except Exception:
        tokenizer = None
    if processor_kwargs is not None and not isinstance(processor_kwargs, dict):      # the reference hands a namedtuple (namedtuple_to_dict, pretrained_model.py:295)
        processor_kwargs = processor_kwargs._asdict() if hasattr(processor_kwargs, '_asdict') else dict(vars(processor_kwargs))
    try:
        processor = AutoProcessor.from_pretrained(path, trust_remote_code=True, **(processor_kwargs or {}))
    except Exception:
        processor = None
    if processor is not None and hasattr(processor, 'tokenizer'):
        processor.tokenizer.padding_side = padding_side
        processor.tokenizer.model_max_length = model_max_length
        return processor.tokenizer, processor
    return tokenizer, None


def load_pretrained(path, device, *, trainable=True, head='lm', dtype=torch.bfloat16, model_max_length=512, padding_side='left',
                    state_from=None, build_kwargs=None, with_tokenizer=True, processor_kwargs=None):
    """Native `load_pretrained_models`: returns (model, tokenizer, processor, hf_config).  `state_from`: another directory to take the
    weights from (same geometry).  The tokenizer gains `<pad>` when it has no pad token and the embeddings grow by that row."""
    from transformers import AutoConfig
    from . import configs
    from .modeling import build_model
    path = os.path.expanduser(str(path))
    hf_config = AutoConfig.from_pretrained(path, trust_remote_code=True)
    tokenizer = processor = None
    extra = 0
    if with_tokenizer:
        tokenizer, processor = load_tokenizer_and_processor(path, model_max_length, padding_side, processor_kwargs)      # pretrained_model.py:183, :295
        if tokenizer is not None:
            if tokenizer.pad_token is None:
                extra = tokenizer.add_special_tokens({'pad_token': DEFAULT_PAD_TOKEN})
            for k in ('bos_token_id', 'eos_token_id', 'pad_token_id'):          # pretrained_model.py:118-120
                setattr(hf_config, k, getattr(tokenizer, k))
    text_cfg = getattr(hf_config, 'text_config', hf_config)
    if extra:
        # model.resize_token_embeddings(len(tokenizer)) (pretrained_model.py:131-146): the embeddings take EXACTLY len(tokenizer) rows -- one more
        # than before for the usual checkpoint, fewer for one whose embedding was padded beyond its tokenizer
        text_cfg.vocab_size = len(tokenizer)
        if text_cfg is not hf_config and hasattr(hf_config, 'vocab_size'):
            hf_config.vocab_size = text_cfg.vocab_size
        if torch.device(device).type == 'cuda':          # (host-side dry runs launch nothing)
            check_vocab_rows(text_cfg.vocab_size, path)
    cfg = configs.from_hf_config(hf_config)
    model = build_model(cfg, device, trainable=trainable, head=head, dtype=dtype, **(build_kwargs or {}))
    sd = LazyCheckpoint(os.path.expanduser(str(state_from)) if state_from else path, cfg['kind'])
    src = _WithNewRows(sd, _embedding_names(sd), extra, rows=text_cfg.vocab_size) if extra else sd
    missing = model.load_state_dict(src, strict=False)
    # tensors of the checkpoint that the native model has no place for: with HF's `strict=False` semantics they would vanish without a word (a
    # bias the config did not announce, an adapter, a second tower) and the model would compute something else than the checkpoint's author ran
    names = model.fuse_expert_keys(src) if hasattr(model, 'fuse_expert_keys') else src
    known = model.store.alias
    benign = lambda k: (k.endswith('rotary_emb.inv_freq')                                   # a buffer old checkpoints carry
                        or (k == 'lm_head.weight' and (head != 'lm' or getattr(model, 'tied', False) or cfg['kind'] == 'opt'))   # unused by a score model / tied
                        or (k.startswith('score_head.') and head == 'lm'))                 # a reward model's head when its backbone is loaded as a language model
    unexpected = [k for k in names if k not in known and not benign(k)]
    if unexpected:
        raise RuntimeError(f'{path}: the checkpoint holds {len(unexpected)} tensors the native {cfg["kind"]} model does not implement, e.g. {unexpected[:4]}')
    # a reward / critic model initialised from a language-model checkpoint has no score head yet (models/reward_model.py): everything else must be there
    bad = [m for m in missing if not m.startswith('score_head')]
    if bad:
        raise RuntimeError(f'{path}: checkpoint lacks {len(bad)} tensors of the native {cfg["kind"]} model, e.g. {bad[:4]}')
    for name in missing:          # a fresh score head: nn.Linear's default initialisation, as AutoModelForScore gives it (models/reward_model.py)
        w = model.store.view(name)
        bound = 1.0 / max(1, w.shape[-1]) ** 0.5
        w.copy_(((torch.rand(tuple(w.shape)) * 2.0 - 1.0) * bound).to(w.dtype))
        for g in model.store.master:
            if model.store.master[g] is not model.store.flat[g]:
                model.store.master[g].copy_(model.store.flat[g])
    sd.close()
    return model, tokenizer, processor, hf_config
