"""Input pipeline either side of the hot path.

The reference's `PreferenceCollator.__call__` (align_anything/datasets/text_image_to_text/preference.py:215-263) runs the
HF processor on the raw conversations and images EVERY step -- tokenisation and image preprocessing are synchronous CPU
work inside the training loop -- and then does a blocking `.to(device)` per tensor.  Here:

* `TokenizedPreferenceCache` runs the processor ONCE per sample (both conversations + the image) and keeps the unpadded
  token ids, the pixel tensor and the response lengths in pinned host memory;
* `CachedPreferenceCollator` builds the batch of the same contract from the cache (rows [0,B) better, [B,2B) worse,
  padded to the longest row on `padding_side`, `pixel_values` = images * 2, `meta_info.response_lens`) -- integer work
  that must equal the reference collator's output exactly;
* `DevicePrefetcher` stages batch k+1 host->HBM on a side HIP stream while step k computes, and pre-builds the response
  window index plan (`trainers/common.py::build_window`) so the step starts with no host work;
* `SupervisedCollator` / `PromptOnlyCollator` are the SFT and rollout counterparts (datasets/text_to_text/supervised.py:139-162,
  prompt_only.py:154-175) on already tokenised samples: pinned host tensors of the same contract, for the same prefetcher
  (which then pre-builds the label window of the supervised loss as well).
"""
from __future__ import annotations

import threading
from queue import Empty, Full, Queue

import torch


def _pin(t: torch.Tensor) -> torch.Tensor:
    return t.pin_memory() if torch.cuda.is_available() and not t.is_pinned() else t


class TokenizedPreferenceCache:
    """samples: the dicts `PreferenceDataset.preprocess` returns (better_conversation, worse_conversation, image,
    better_response_lens, worse_response_lens; preference.py:132-160).  processor(text=..., images=..., return_tensors='pt')
    is called once per conversation, without padding."""

    def __init__(self, samples, processor, has_images: bool = True, processor_kwargs: dict | None = None):
        """processor_kwargs: extra arguments of the per-conversation call, to mirror the collator being replaced -- the text-to-text
        PreferenceCollator tokenises with `add_special_tokens=False` (datasets/text_to_text/preference.py:186-193), the text+image one
        with the processor's defaults (datasets/text_image_to_text/preference.py:232-239)."""
        self.items = []
        extra = dict(processor_kwargs or {})
        for s in samples:
            img = s.get('image') if has_images else None
            kw = dict(extra, images=img) if img is not None else dict(extra)
            b = processor(text=s['better_conversation'], return_tensors='pt', **kw)
            w = processor(text=s['worse_conversation'], return_tensors='pt', **kw)
            item = {'better_ids': _pin(b['input_ids'][0].to(torch.int64).contiguous()),
                    'worse_ids': _pin(w['input_ids'][0].to(torch.int64).contiguous()),
                    'better_response_lens': int(s['better_response_lens']), 'worse_response_lens': int(s['worse_response_lens'])}
            if 'pixel_values' in b:
                item['pixel_values'] = _pin(b['pixel_values'][0].contiguous())
            self.items.append(item)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


class CachedPreferenceCollator:
    def __init__(self, pad_token_id: int, padding_side: str = 'left'):
        if padding_side not in ('left', 'right'):
            raise ValueError(f'padding_side must be left or right, got {padding_side!r}')
        self.pad_token_id, self.padding_side = int(pad_token_id), padding_side

    def __call__(self, items) -> dict:
        rows = [it['better_ids'] for it in items] + [it['worse_ids'] for it in items]
        T = max(int(r.numel()) for r in rows)
        ids = torch.full((len(rows), T), self.pad_token_id, dtype=torch.int64)
        mask = torch.zeros((len(rows), T), dtype=torch.int64)
        for r, row in enumerate(rows):
            n = int(row.numel())
            if self.padding_side == 'left':
                ids[r, T - n:] = row; mask[r, T - n:] = 1
            else:
                ids[r, :n] = row; mask[r, :n] = 1
        batch = {'input_ids': _pin(ids), 'attention_mask': _pin(mask),
                 'meta_info': {'response_lens': [it['better_response_lens'] for it in items] + [it['worse_response_lens'] for it in items]}}
        if self.padding_side == 'left':
            # host integers for the opt-in shared-prompt packing (trainers/common.py::build_pack_plan): known here for free, a device read there
            n = len(items)
            shared = []
            for a, b in zip(rows[:n], rows[n:]):
                m = min(int(a.numel()), int(b.numel()))
                ne = (a[:m] != b[:m]).nonzero()
                shared.append(int(ne[0]) if ne.numel() else m)
            batch['meta_info']['seq_lens'] = [int(r.numel()) for r in rows]
            batch['meta_info']['shared_prefix_lens'] = shared
        if 'pixel_values' in items[0]:
            pv = torch.stack([it['pixel_values'] for it in items])
            batch['pixel_values'] = _pin(torch.cat([pv, pv], 0))      # images * 2 (preference.py:219-222)
        return batch


def _pad(rows, value, side, dtype=torch.int64):
    T = max(int(r.numel()) for r in rows)
    out = torch.full((len(rows), T), value, dtype=dtype)
    for r, row in enumerate(rows):
        n = int(row.numel())
        if side == 'left':
            out[r, T - n:] = row
        else:
            out[r, :n] = row
    return out


class SupervisedCollator:
    """datasets/text_to_text/supervised.py:139-162: input_ids right-padded with pad_token_id, labels right-padded with -100,
    attention_mask = input_ids != pad_token_id (a bool tensor, as the reference returns).  samples: dicts with 1-D `input_ids` / `labels`."""

    IGNORE_INDEX = -100

    def __init__(self, pad_token_id: int):
        self.pad_token_id = int(pad_token_id)

    def __call__(self, samples) -> dict:
        ids = _pad([s['input_ids'] for s in samples], self.pad_token_id, 'right')
        labels = _pad([s['labels'] for s in samples], self.IGNORE_INDEX, 'right')
        return {'input_ids': _pin(ids), 'labels': _pin(labels), 'attention_mask': _pin(ids.ne(self.pad_token_id))}


class UnmatchedSupervisedCollator:
    """datasets/text_to_text/supervised.py:196-219, the batches KTO estimates its KL term on (a prompt paired with a neighbour's response):
    input_ids right-padded, attention_mask = ids != pad, labels None, meta_info.response_lens from the samples."""

    def __init__(self, pad_token_id: int):
        self.pad_token_id = int(pad_token_id)

    def __call__(self, samples) -> dict:
        ids = _pad([s['input_ids'] for s in samples], self.pad_token_id, 'right')
        return {'input_ids': _pin(ids), 'labels': None, 'attention_mask': _pin(ids.ne(self.pad_token_id)),
                'meta_info': {'response_lens': [int(s['response_lens']) for s in samples]}}


class PromptOnlyCollator:
    """datasets/text_to_text/prompt_only.py:154-175: prompts LEFT-padded with pad_token_id; the mask marks the real tokens of every
    row (all of them, also a pad id inside the text) -- what `generate` and the PPO rollout consume."""

    def __init__(self, pad_token_id: int):
        self.pad_token_id = int(pad_token_id)

    def __call__(self, samples) -> dict:
        rows = [s['input_ids'] for s in samples]
        ids = _pad(rows, self.pad_token_id, 'left')
        mask = _pad([torch.ones(int(r.numel()), dtype=torch.bool) for r in rows], False, 'left', dtype=torch.bool)
        return {'input_ids': _pin(ids), 'attention_mask': _pin(mask)}


# Host-side fetches (dataset __getitem__ + collator: tokenizer and image-processor calls) of ALL prefetchers are serialised: a mid-epoch `eval()`
# iterates the evaluation loader while the training loader's producer is a batch ahead, and both datasets hold the same fast tokenizer, which
# is not re-entrant ("Already borrowed").  The reference loads single-threaded (num_workers = 0), so nothing is lost against it.
_HOST_FETCH = threading.Lock()


class DevicePrefetcher:
    """Iterates a host dataloader one batch ahead: the next batch's tensors are copied to the device on a side stream
    (non_blocking from pinned memory) and its window plan is built, while the current step runs.  `pad_token_id` given
    -> `_window` is attached (DPOTrainer._window then finds it)."""

    def __init__(self, loader, device, pad_token_id=None, depth: int = 2):
        self.loader, self.device, self.pad_token_id, self.depth = loader, torch.device(device), pad_token_id, max(1, depth)
        self.stream = torch.cuda.Stream(self.device) if self.device.type == 'cuda' else None

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        out = {}
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                for k, v in batch.items():
                    out[k] = v.to(self.device, non_blocking=True) if isinstance(v, torch.Tensor) else v
                if isinstance(batch.get('labels'), torch.Tensor):
                    out['_labels_host'] = batch['labels']          # the label plan is integer host work: keep the host copy for it
                ev = torch.cuda.Event()
                ev.record(self.stream)
            out['_ready'] = ev
        else:
            out = dict(batch)
            if isinstance(batch.get('labels'), torch.Tensor):
                out['_labels_host'] = batch['labels']
        return out

    def _finish(self, out):
        ev = out.pop('_ready', None)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        if self.pad_token_id is not None and 'meta_info' in out and 'response_lens' in out['meta_info']:
            from .trainers.common import build_window
            out['_window'] = build_window(out['input_ids'], out['meta_info']['response_lens'], self.pad_token_id)
        if out.get('labels') is not None and out.get('_labels_host') is not None:
            from .trainers.common import build_label_window
            out['_window'] = build_label_window(out.pop('_labels_host'), device=self.device)     # supervised loss rows (trainers/sft.py)
        return out

    def _retire(self):
        """Stop the producer of an iterator that was abandoned before its end (`next(iter(loader))`, a `break`, an exception in the step):
        it would otherwise sit in `q.put` forever holding `depth` staged batches, or -- worse -- keep running the dataset's tokenizer next
        to the producer of the NEXT iterator (a fast tokenizer is not re-entrant: "Already borrowed")."""
        prev = getattr(self, '_live', None)
        if prev is None:
            return
        cancel, q, t = prev
        cancel.set()
        while t.is_alive():
            try:
                q.get(timeout=0.05)         # unblock a producer waiting on a full queue
            except Empty:
                pass
        self._live = None

    def __iter__(self):
        self._retire()
        q: Queue = Queue(maxsize=self.depth)
        stop = object()
        cancel = threading.Event()

        def put(item) -> bool:
            while not cancel.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except Full:
                    continue
            return False

        def producer():
            try:
                it = iter(self.loader)
                while True:
                    with _HOST_FETCH:        # one prefetcher inside a dataset at a time: the train and the eval loader share ONE tokenizer
                        try:
                            b = next(it)
                        except StopIteration:
                            break
                    if not put(self._stage(b)):
                        return
            except BaseException as ex:   # surface loader errors in the consumer
                put(ex)
            put(stop)

        t = threading.Thread(target=producer, daemon=True)
        self._live = (cancel, q, t)
        t.start()
        try:
            while True:
                try:
                    item = q.get(timeout=0.2)
                except Empty:
                    if cancel.is_set():       # a newer iteration of the same loader retired this one's producer: fail, do not wait forever
                        raise RuntimeError('DevicePrefetcher: this iterator was retired by a newer iteration of the same loader') from None
                    continue
                if item is stop:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield self._finish(item)
            t.join()
            if getattr(self, '_live', None) is not None and self._live[2] is t:
                self._live = None
        finally:
            if getattr(self, '_live', None) is not None and self._live[2] is t:
                self._retire()
