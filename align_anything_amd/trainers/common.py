"""Host-side plumbing shared by the native trainers: response-window index building (integer work on
python ints that the collator already holds on the host), metric reduction, config access."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .. import ops


def pad64(n: int) -> int:
    return max(64, (n + 63) // 64 * 64)


def build_window(input_ids: torch.Tensor, response_lens, pad_token_id: int):
    """Index plan for `logits[idx][-R:][:-1]` x `strip_pad(ids[idx])[-R:][1:]`
    (align_anything/trainers/text_to_text/dpo.py:131-139).  response_lens are host ints
    (batch['meta_info']['response_lens']), so the positional part is built on the CPU with no device sync; the
    label part depends on where pad ids sit and is computed on the device (aa_window_labels)."""
    N, T = input_ids.shape
    dev = input_ids.device
    R = np.asarray([int(r) for r in response_lens], dtype=np.int64)
    if len(R) != N:
        raise ValueError(f'response_lens has {len(R)} entries for {N} rows')
    if (R < 1).any() or (R > T).any():
        raise ValueError(f'response_lens must be within [1, {T}]: {R.tolist()}')
    cnt = R - 1
    off = np.concatenate([[0], np.cumsum(cnt)])
    rows = int(off[-1])
    rows_pad = pad64(rows)
    Mp = (N * T + 63) // 64 * 64
    row_idx = np.zeros(rows_pad, dtype=np.int64)
    seq_of_row = np.zeros(rows, dtype=np.int64)
    col_of_row = np.zeros(rows, dtype=np.int64)
    for n in range(N):
        j = np.arange(cnt[n])
        row_idx[off[n]:off[n + 1]] = n * T + (T - R[n]) + j
        seq_of_row[off[n]:off[n + 1]] = n
        col_of_row[off[n]:off[n + 1]] = j
    inv = np.full(Mp, -1, dtype=np.int32)
    inv[row_idx[:rows]] = np.arange(rows, dtype=np.int32)
    w = {
        'N': N, 'T': T, 'rows': rows, 'rows_pad': rows_pad, 'max_len': int(cnt.max()) if N else 0,
        'row_idx': torch.from_numpy(row_idx).to(dev, non_blocking=True),
        'inv_map': torch.from_numpy(inv).to(dev, non_blocking=True),
        'seq_off': torch.from_numpy(off.astype(np.int32)).to(dev, non_blocking=True),
        'flat_to_padded': torch.from_numpy(seq_of_row * max(int(cnt.max()), 1) + col_of_row).to(dev, non_blocking=True),
        'resp_len': torch.from_numpy(R.astype(np.int32)).to(dev, non_blocking=True),
        'row_off': torch.from_numpy(off[:-1].astype(np.int32)).to(dev, non_blocking=True),
        # for the dead-row elimination of the last decoder layer (modeling.LlamaStack.forward `tail`): the window rows as an identity gather, the first query
        # position anybody consumes per sequence, and the share of the causal attention work that remains (host ints: the bench's FLOP count)
        **tail_fields(row_idx, rows, rows_pad, N, T, dev),
    }
    labels = torch.zeros(rows_pad, dtype=torch.int64, device=dev)
    ops.window_labels(input_ids.contiguous(), pad_token_id, w['resp_len'], w['row_off'], labels)
    w['labels'] = labels
    return w


def tail_fields(row_idx: np.ndarray, rows: int, rows_pad: int, N: int, T: int, dev) -> dict:
    """What the dead-row elimination of the last decoder layer (modeling.LlamaStack.forward `tail`) needs from a window: the window rows as an identity
    gather, per sequence the first query position anybody consumes (T: none) and the share of the causal attention work that remains.  Host arithmetic."""
    q = np.full(N, T, dtype=np.int64)
    if rows:
        r = row_idx[:rows]
        np.minimum.at(q, r // T, r % T)
    # the gather index of the window rows with its PAD entries pointing at the first window row, not at row 0: the attention output of a skipped query block is
    # never written (uninitialised memory, possibly NaN), and a pad row that copied it would meet its exactly-zero gradient as 0 x NaN in the weight gradients
    # (and send a NaN through the MoE router)
    ga = row_idx.copy()
    ga[rows:] = row_idx[0] if rows else 0
    return {'row_idx_id': torch.arange(rows_pad, dtype=torch.int64, device=dev),
            'tail_gather': torch.from_numpy(ga).to(dev, non_blocking=True),
            'tail_qskip': torch.from_numpy(q.astype(np.int32)).to(dev, non_blocking=True),
            'tail_frac': float(np.mean(1.0 - (q // 64 * 64 / T) ** 2)) if N else 1.0}


def build_pack_plan(input_ids: torch.Tensor, attention_mask, window, meta_info=None):
    """Shared-prompt packing (round 6; opt-in: train_cfgs.share_prompt_prefix).  The chosen and the rejected row of a preference pair start with the SAME
    tokens -- BOS, the image tokens, the prompt (datasets/text_image_to_text/preference.py:132-160 formats both conversations from one prompt) -- and under a
    causal mask the hidden states of those positions do not depend on what follows them: the reference computes them twice (rows [0, B) and [B, 2B) of
    `model(**batch)`, trainers/text_image_to_text/dpo.py:85-105).  Here every row-wise operation of the decoder (norms, the four projections, SwiGLU,
    residuals, embedding, lm_head window) runs on a PACKED token set -- per pair: the shared prefix once, then the rest of the chosen and of the rejected
    sequence -- and only attention, the one operation that mixes positions, runs on the reference's [2B, T] layout, fed by a row gather of the packed
    q | k | v and followed by a row gather back.  At T = 2048, R = 512 that is 62.5 % of the token rows: the GEMMs, 95 % of the step's FLOPs, shrink with it.

    Exactness.  A packed row's value is the value the unpacked computation gives the chosen row's copy of that position (row-wise kernels do not see the
    other rows; the attention kernel sees the reference layout) -- bit for bit in the forward pass when both rows carry the same left padding.  With
    different response lengths the collator pads the two rows differently; HF numbers positions by slot (SURVEY 8(a'): `arange(T)` counts the pads), so the
    two copies of the prefix sit at rotary positions that differ by delta = pad_r - pad_c.  The packed computation gives BOTH rows of a pair one frame, the
    longer row's (token j of either sequence at position min(pad_c, pad_r) + j, always < T): the shorter row is evaluated delta positions to the left of where
    HF puts it, all RELATIVE positions, hence all attention scores, are those of the reference up to the rounding of the rotary tables at another absolute
    position (SURVEY measured 1e-7 in fp32; against the reference trainer at full depth: profiles/parity/parity_llava7b_full_depth_packed_vs_reference.txt).  In the backward pass the gradient of a shared row is the
    sum of its two copies' gradients, formed in the activation dtype before the weight-gradient GEMM instead of inside its fp32 accumulator.

    Returns None when nothing can be shared (no pair has a common prefix of >= 64 tokens).  Host integers come from `meta_info` when the collator provides
    them (`seq_lens`, `shared_prefix_lens`: no device read); otherwise they are read from the batch (one device -> host sync per batch)."""
    N, T = input_ids.shape
    if N % 2:
        return None
    B = N // 2
    meta_info = meta_info or {}
    R = np.asarray([int(r) for r in meta_info['response_lens']], dtype=np.int64)
    if 'seq_lens' in meta_info:
        lens = np.asarray([int(x) for x in meta_info['seq_lens']], dtype=np.int64)
    elif attention_mask is None:
        lens = np.full(N, T, dtype=np.int64)
    else:
        lens = attention_mask.sum(1).cpu().numpy().astype(np.int64)
    pad = T - lens
    if 'shared_prefix_lens' in meta_info:
        Lp = np.asarray([int(x) for x in meta_info['shared_prefix_lens']], dtype=np.int64)
    else:
        ids_h = input_ids.cpu().numpy()
        Lp = np.zeros(B, dtype=np.int64)
        for i in range(B):
            a, b = ids_h[i, pad[i]:], ids_h[B + i, pad[B + i]:]
            m = min(len(a), len(b))
            ne = np.nonzero(a[:m] != b[:m])[0]
            Lp[i] = int(ne[0]) if len(ne) else m
    Lp = np.minimum(Lp, np.minimum(lens[:B] - R[:B], lens[B:] - R[B:]))          # the prompt only: every response-window row stays a row of its own
    Lp = np.where(Lp >= 64, Lp, 0)
    if not Lp.any() and int(pad.sum()) * 8 < N * T:
        return None          # nothing to share and less than 1/8 of the slots are padding: the reference layout as it is
    # (nothing to share but a lot of padding: the plan still drops the pad slots from every row-wise kernel -- each row keeps its own positions, bit-identical)
    Mp = (N * T + 63) // 64 * 64
    slot2row = np.full(Mp, -1, dtype=np.int32)          # full slot (n, t) -> packed row (-1: a pad slot)
    owner = np.full(Mp, -1, dtype=np.int32)             # ... only where that slot OWNS the row (the rejected copy of a shared prefix does not)
    rows = int((lens[:B] + lens[B:] - Lp).sum())
    Mq = (rows + 63) // 64 * 64
    row2slot = np.full(Mq, -1, dtype=np.int32)          # packed row -> owning slot
    row2slot_b = np.full(Mq, -1, dtype=np.int32)        # packed row -> the second slot that holds a copy of it (shared prefix rows), else -1
    pos = np.zeros(Mq, dtype=np.int32)
    qskip = np.zeros(N, dtype=np.int32)                # per row of the reference layout: first query slot anyone consumes (attention skips whole blocks below)
    r0 = 0
    for i in range(B):
        c, rj = i, B + i
        lp, lc, lr = int(Lp[i]), int(lens[c]), int(lens[rj])
        sc = c * T + pad[c] + np.arange(lc)              # chosen row: slots of its tokens
        sr = rj * T + pad[rj] + np.arange(lr)
        rows_c = r0 + np.arange(lc)                      # prefix + chosen remainder, in sequence order
        rows_r = np.concatenate([r0 + np.arange(lp), r0 + lc + np.arange(lr - lp)])
        slot2row[sc] = rows_c; owner[sc] = rows_c
        slot2row[sr] = rows_r; owner[sr[lp:]] = rows_r[lp:]
        row2slot[rows_c] = sc
        row2slot[rows_r[lp:]] = sr[lp:]
        row2slot_b[rows_r[:lp]] = sr[:lp]
        qskip[rj] = pad[rj] + lp                         # the rejected row's copy of the prefix: its outputs are taken from the chosen row
        f0 = min(pad[c], pad[rj])                        # HF: position = slot index within the row.  The pair's frame is the LONGER row's (smaller left pad):
        fc, fr = (f0, f0) if lp else (pad[c], pad[rj])   # that row keeps its own positions, the other moves by |pad_r - pad_c|, and every position stays < T
        pos[rows_c] = fc + np.arange(lc)                 # (no shared prefix: no common frame is needed, both rows keep their own)
        pos[rows_r[lp:]] = fr + lp + np.arange(lr - lp)
        r0 += lc + lr - lp
    assert r0 == rows
    dev = input_ids.device
    up = lambda a: torch.from_numpy(a).to(dev, non_blocking=True)
    s2r = up(slot2row)
    skipped = (qskip // 256 * 256).astype(np.float64)      # whole 256-row query blocks below qskip: the part of the causal triangle attention leaves out
    attn_frac = float(1.0 - (skipped ** 2).sum() / (N * float(T) ** 2))
    plan = {'N': N, 'T': T, 'Mq': Mq, 'rows': rows, 'full_rows': N * T, 'slot2row': s2r, 'owner': up(owner), 'row2slot': up(row2slot),
            'row2slot_b': up(row2slot_b), 'pos': up(pos), 'qskip': up(qskip), 'attn_frac': attn_frac, 'shared_rows': int(Lp.sum()), 'prefix_lens': Lp.tolist()}
    # packed token ids (pad rows: id 0, never an image token) and rotary positions of the FULL layout in the packed frame (the attention backward's epilogue)
    ids_full = input_ids.reshape(-1)
    r2s = plan['row2slot'].long().clamp(min=0)
    plan['ids'] = torch.where(plan['row2slot'] >= 0, ids_full[r2s.clamp(max=N * T - 1)], torch.zeros_like(ids_full[:1]))
    plan['pos_full'] = torch.where(s2r >= 0, plan['pos'][s2r.long().clamp(min=0)], torch.zeros_like(plan['pos'][:1]))
    # the response window in packed rows
    w = dict(window)
    ri = window['row_idx']
    w['row_idx'] = s2r[ri].long().clamp(min=0)
    inv = np.full(Mq, -1, dtype=np.int32)
    plan['_inv_host'] = inv          # filled on the device below (row_idx lives there)
    inv_t = torch.full((Mq,), -1, dtype=torch.int32, device=dev)
    inv_t[w['row_idx'][:window['rows']]] = torch.arange(window['rows'], dtype=torch.int32, device=dev)
    w['inv_map'] = inv_t
    plan['window'] = w
    if 'tail_qskip' in window:       # the last layer on the window rows only (modeling.LlamaStack.forward `tail`): attention rows by [N, T] slot, stack rows by packed row
        ga = window['tail_gather']
        plan['tail'] = {'gather_attn': ga, 'gather_x': s2r[ga].long().clamp(min=0), 'scatter_attn': window['inv_map'], 'scatter_x': inv_t,
                        'qskip': window['tail_qskip'], 'frac': window['tail_frac']}
    del plan['_inv_host']
    return plan


def flat_to_padded(flat_logp: torch.Tensor, w) -> torch.Tensor:
    """pad_sequence(..., padding_value=0.0) layout of dpo.py:140-142: [N, max(R)-1], right padded with 0."""
    L = max(w['max_len'], 1)
    out = torch.zeros(w['N'] * L, dtype=flat_logp.dtype, device=flat_logp.device)
    out[w['flat_to_padded']] = flat_logp[:w['rows']]
    return out.view(w['N'], L)[:, :w['max_len']]


def get_all_reduce_mean(t: torch.Tensor) -> torch.Tensor:
    """align_anything/utils/multi_process.py:74-82."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.AVG if t.is_cuda else dist.ReduceOp.SUM)
        if not t.is_cuda:
            t /= dist.get_world_size()
    return t


def get_all_reduce_max(t: torch.Tensor) -> torch.Tensor:
    """align_anything/utils/multi_process.py:85-89."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


END_AT_LAST_POSITION_KINDS = ('llava', 'qwen2vl')


def end_index(kind: str, attention_mask: torch.Tensor) -> torch.Tensor:
    """Position whose score is the reward model's `end_scores`, per backbone as the reference's score models define it:
    the LAST ATTENDED token for OPT / Llama / Qwen2-Audio / Qwen3-MoE (models/opt.py:67, llama.py:71, qwen2_audio.py:80,
    qwen3_moe.py:71: `attention_mask[i].nonzero()[-1]`), but position -1 REGARDLESS of the mask for the vision-language reward
    models (models/llava.py:64-68, qwen2_vl.py:61-64: `last_hidden_state[:, -1]`).  The two agree on left-padded PPO
    experience; they differ on the right-padded batches of reward-model training (trainers/text_to_text/rm.py:81)."""
    N, T = attention_mask.shape
    if kind in END_AT_LAST_POSITION_KINDS:
        return torch.full((N,), T - 1, dtype=torch.int64, device=attention_mask.device)
    return (attention_mask.to(torch.int64) * torch.arange(1, T + 1, device=attention_mask.device)[None]).argmax(dim=1)


def cfg_get(cfgs, path: str, default=None):
    """Read `a.b.c` from a namedtuple / namespace / dict config (missing -> default), mirroring the reference's
    dict_to_namedtuple objects whose missing attributes read as None (utils/tools.py:87-93)."""
    cur = cfgs
    for part in path.split('.'):
        if cur is None:
            return default
        cur = cur.get(part) if isinstance(cur, dict) else getattr(cur, part, None)
    return default if cur is None else cur


def resume_from_slice(trainer, engine_attr: str = 'model') -> None:
    """`train_cfgs.load_checkpoint` (supervised_trainer.py:76-77, :267-268): `model_cfgs.model_name_or_path` is a `slice_<step>` directory written with
    `save_checkpoint` -- the step counter continues at <step> and the engine takes its fp32 masters, Adam moments and update count from the slice
    (`engine.load_checkpoint`), after the 16-bit weights were loaded from the same directory like any other checkpoint."""
    if not cfg_get(trainer.cfgs, 'train_cfgs.load_checkpoint', False):
        return
    path = str(cfg_get(trainer.cfgs, 'model_cfgs.model_name_or_path', '') or '')
    try:
        trainer.global_step = int(path.rstrip('/').split('slice_')[-1])
    except ValueError:
        raise ValueError(f'train_cfgs.load_checkpoint: model_cfgs.model_name_or_path must be a slice_<step> directory, got {path!r}') from None
    getattr(trainer, engine_attr).load_checkpoint(load_dir=path)


def refuse_unsupported_options(cfgs) -> None:
    """Options of the reference's yaml that change WHAT is trained and have no native implementation must stop the trainer, not be dropped:
    `lora_cfgs.use_lora` (base/supervised_trainer.py:53-58: peft adapters instead of full fine-tuning), `bnb_cfgs.use_bnb` (4 / 8-bit weights) and
    `train_cfgs.fp16` (DeepSpeed fp16 with loss scaling)."""
    if cfg_get(cfgs, 'lora_cfgs.use_lora', False):
        raise NotImplementedError('lora_cfgs.use_lora: LoRA adapters have no native implementation (the native trainers fine-tune the full weights)')
    if cfg_get(cfgs, 'bnb_cfgs.use_bnb', False):
        raise NotImplementedError('bnb_cfgs.use_bnb: quantised base weights have no native implementation')
    if cfg_get(cfgs, 'train_cfgs.fp16', False):
        raise NotImplementedError('train_cfgs.fp16: fp16 training with loss scaling has no native implementation (bf16, the reference\'s default, and the fp32 parity mode are built)')


def build_span_window(input_ids: torch.Tensor, start: int):
    """Rows for `gather_log_probabilities(logits[:, :-1], input_ids[:, 1:])[:, start:]` and
    `scores[:, :-1][:, start:]` (align_anything/trainers/text_to_text/ppo.py:339-356): row (n, j) for
    j in [start, T-2] reads hidden position j and is labelled with token j+1.  Pure positional -> host built."""
    N, T = input_ids.shape
    dev = input_ids.device
    W = T - 1 - start
    if W < 1:
        raise ValueError(f'prompt_idx {start} leaves no response positions in a length-{T} sequence')
    rows = N * W
    rows_pad = pad64(rows)
    Mp = (N * T + 63) // 64 * 64
    j = np.arange(start, T - 1)
    row_idx = np.zeros(rows_pad, dtype=np.int64)
    row_idx[:rows] = (np.arange(N)[:, None] * T + j[None, :]).reshape(-1)
    inv = np.full(Mp, -1, dtype=np.int32)
    inv[row_idx[:rows]] = np.arange(rows, dtype=np.int32)
    labels = torch.zeros(rows_pad, dtype=torch.int64, device=dev)
    labels[:rows] = input_ids[:, start + 1:].reshape(-1)
    return {'N': N, 'T': T, 'W': W, 'rows': rows, 'rows_pad': rows_pad,
            'row_idx': torch.from_numpy(row_idx).to(dev, non_blocking=True),
            'inv_map': torch.from_numpy(inv).to(dev, non_blocking=True), 'labels': labels, **tail_fields(row_idx, rows, rows_pad, N, T, dev)}


def build_tail_window(input_ids: torch.Tensor, response_lens):
    """Rows of `logits[idx, :-1][-R:]` x `input_ids[idx, 1:][-R:]` (align_anything/trainers/text_image_to_text/ppo.py:233-241,
    302-305; also `scores[:, :-1][idx][-R:]`): row (n, j), j in [T-1-R_n, T-2], reads hidden position j and is labelled with
    token j+1 -- R_n entries per sequence, flat, in the layout of build_window.  Pure positional: built on the host."""
    N, T = input_ids.shape
    dev = input_ids.device
    R = np.asarray([int(r) for r in response_lens], dtype=np.int64)
    if len(R) != N or (R < 1).any() or (R > T - 1).any():
        raise ValueError(f'response_lens must have {N} entries within [1, {T - 1}]: {R.tolist()}')
    off = np.concatenate([[0], np.cumsum(R)])
    rows, rows_pad, L = int(off[-1]), pad64(int(off[-1])), int(R.max())
    Mp = (N * T + 63) // 64 * 64
    row_idx = np.zeros(rows_pad, dtype=np.int64)
    f2p = np.zeros(rows, dtype=np.int64)
    for n in range(N):
        j = np.arange(R[n])
        row_idx[off[n]:off[n + 1]] = n * T + (T - 1 - R[n]) + j
        f2p[off[n]:off[n + 1]] = n * L + j
    inv = np.full(Mp, -1, dtype=np.int32)
    inv[row_idx[:rows]] = np.arange(rows, dtype=np.int32)
    ridx = torch.from_numpy(row_idx).to(dev, non_blocking=True)
    labels = torch.zeros(rows_pad, dtype=torch.int64, device=dev)
    labels[:rows] = input_ids.reshape(-1)[ridx[:rows] + 1]
    return {'N': N, 'T': T, 'rows': rows, 'rows_pad': rows_pad, 'max_len': L, 'row_idx': ridx,
            'inv_map': torch.from_numpy(inv).to(dev, non_blocking=True),
            'seq_off': torch.from_numpy(off.astype(np.int32)).to(dev, non_blocking=True),
            'flat_to_padded': torch.from_numpy(f2p).to(dev, non_blocking=True), 'labels': labels, **tail_fields(row_idx, rows, rows_pad, N, T, dev)}


def build_label_window(labels: torch.Tensor, ignore_index: int = -100, device=None):
    """Rows of the supervised loss (hf:loss/loss_utils.py ForCausalLMLoss on datasets/text_to_text/supervised.py:96-99 labels):
    hidden position j of row n predicts labels[n, j + 1]; every (n, j) with j < T - 1 and labels[n, j + 1] != ignore_index is one
    row, in row-major order.  The labels come from the collator (host tensors): built on the host, like the other plans, and
    placed on `device` (default: where the labels live -- pass the model's device for host-side labels)."""
    lab = labels.detach().cpu().numpy()
    N, T = lab.shape
    tgt = lab[:, 1:]
    n_i, j_i = np.nonzero(tgt != ignore_index)
    rows = int(n_i.size)
    if rows == 0:
        raise ValueError('supervised batch without a single label position (all labels == ignore_index)')
    rows_pad = pad64(rows)
    Mp = (N * T + 63) // 64 * 64
    row_idx = np.zeros(rows_pad, dtype=np.int64)
    row_idx[:rows] = n_i * T + j_i
    inv = np.full(Mp, -1, dtype=np.int32)
    inv[row_idx[:rows]] = np.arange(rows, dtype=np.int32)
    lbl = np.zeros(rows_pad, dtype=np.int64)
    lbl[:rows] = tgt[n_i, j_i]
    dev = labels.device if device is None else torch.device(device)
    return {'N': N, 'T': T, 'rows': rows, 'rows_pad': rows_pad, 'row_idx': torch.from_numpy(row_idx).to(dev, non_blocking=True),
            'inv_map': torch.from_numpy(inv).to(dev, non_blocking=True), 'labels': torch.from_numpy(lbl).to(dev, non_blocking=True),
            **tail_fields(row_idx, rows, rows_pad, N, T, dev)}


def pad_rows(flat_2d: torch.Tensor, rows_pad: int) -> torch.Tensor:
    """[B, W] gradient -> flat fp32 [rows_pad] with zero tail (the pad rows of a window carry no gradient)."""
    out = torch.zeros(rows_pad, dtype=torch.float32, device=flat_2d.device)
    out[:flat_2d.numel()] = flat_2d.reshape(-1)
    return out


def compute_dtype(name):
    """'bf16' | 'fp32' (or a torch dtype) -> torch dtype of weights/activations on the native path."""
    if isinstance(name, torch.dtype):
        if name in (torch.bfloat16, torch.float32):
            return name
        raise ValueError(f'compute dtype {name} not supported (bfloat16, float32)')
    table = {'bf16': torch.bfloat16, 'bfloat16': torch.bfloat16, 'fp32': torch.float32, 'float32': torch.float32}
    if str(name).lower() not in table:
        raise ValueError(f'compute dtype {name!r} not supported (bf16, fp32)')
    return table[str(name).lower()]


def expert_parallel_kwargs(cfgs, *model_cfgs) -> dict:
    """`train_cfgs.expert_parallel: true` on a sparse-MoE backbone: the experts are split over the data-parallel ranks
    (expert_parallel.py).  Returns the `build_model` keyword for every model of the trainer -- ONE ExpertParallel on a communicator of
    its own (the token exchange never queues behind a gradient bucket), shared by the trainer's models, which run one after the other in
    the same order on every rank -- or {} when the flag is off / the backbones are dense.  The exchange is the capacity-padded, host-sync-free
    one unless `train_cfgs.expert_parallel_capacity_factor: 0`."""
    if not bool(cfg_get(cfgs, 'train_cfgs.expert_parallel', False)) or not any(c is not None and c.get('kind') == 'qwen3moe' for c in model_cfgs):
        return {}
    import torch.distributed as dist
    from ..expert_parallel import ExpertParallel
    # train_cfgs.expert_parallel_capacity_factor (native key, default 2.0): rows per peer block of the sync-free exchange as a multiple of the
    # balanced share; 0 = the exact exchange (one host read per MoE block and direction)
    return {'ep': ExpertParallel(dist.new_group(), capacity_factor=float(cfg_get(cfgs, 'train_cfgs.expert_parallel_capacity_factor', 2.0)),
                                 dense_below=int(cfg_get(cfgs, 'train_cfgs.expert_parallel_dense_below', 4096)))}


def save_slice(trainer, engine, tag=None, output_dir=None) -> str:
    """base/supervised_trainer.py:404-450 layout: <output_dir>/slice_<tag|end>/{config.json, tokenizer / processor files,
    pytorch_model.bin} -- a directory `AnyModel.from_pretrained` loads.  The weights come from the native engine under their HF names;
    config.json / tokenizer / processor are written when the trainer was handed the HF objects (`hf_config`, `tokenizer`, `processor`
    attributes: the native path itself only needs the plain-dict geometry)."""
    import os
    out = output_dir or cfg_get(trainer.cfgs, 'logger_cfgs.output_dir', './output')
    d = os.path.join(out, f'slice_{tag or "end"}')
    engine.save_16bit_model(d, save_filename='pytorch_model.bin')          # every rank calls it (expert shards are gathered), rank 0 writes
    # supervised_trainer.py:435-436 / rl_trainer.py:362-363: `train_cfgs.save_checkpoint` adds the engine's own training state (fp32 masters, Adam
    # moments, step count -- DeepSpeed's save_checkpoint) to the slice, which `train_cfgs.load_checkpoint` resumes from
    if cfg_get(trainer.cfgs, 'train_cfgs.save_checkpoint', False) and getattr(engine, 'trainable', False):
        engine.save_checkpoint(d)
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0:
        for obj in (getattr(trainer, 'hf_config', None), getattr(trainer, 'tokenizer', None), getattr(trainer, 'processor', None)):
            if obj is not None and hasattr(obj, 'save_pretrained'):
                obj.save_pretrained(d)
    return d


def save_interval(cfgs, total_steps) -> int:
    """`total // logger_cfgs.save_total_limit` of the reference's loops (dpo.py:285-293, rm.py:307-314, ppo.py:462-468, grpo.py:368-375);
    0 = no periodic saves (no limit configured, or the total is unknown)."""
    limit = cfg_get(cfgs, 'logger_cfgs.save_total_limit', None)
    if not limit or not total_steps:
        return 0
    return max(1, int(total_steps) // int(limit))


def eval_due(cfgs, when: str, global_step: int = 0) -> bool:
    """When the reference's training loops call `self.eval()` (text_to_text/rm.py:274-325, ppo.py:422-485, grpo.py:345-384): only with
    `data_cfgs.eval_datasets` set; 'begin' = before the first step; 'steps' = `train_cfgs.eval_strategy == 'steps'` and global_step a multiple of
    `train_cfgs.eval_interval`; 'epoch' = `eval_strategy == 'epoch'` at an epoch's end (PPO; the RM loop evaluates after EVERY epoch whatever the
    strategy -- its caller passes when='always')."""
    if not cfg_get(cfgs, 'data_cfgs.eval_datasets', None):
        return False
    if when in ('begin', 'always'):
        return True
    strategy = cfg_get(cfgs, 'train_cfgs.eval_strategy', 'epoch')
    if when == 'steps':
        every = int(cfg_get(cfgs, 'train_cfgs.eval_interval', 0) or 0)
        return strategy == 'steps' and every > 0 and global_step % every == 0
    return when == 'epoch' and strategy == 'epoch'


@torch.no_grad()
def rl_eval(trainer, eval_dataloader=None) -> dict:
    """`RLTrainerBase.eval` (base/rl_trainer.py:289-329): sample a completion for every prompt of the evaluation set with the ACTOR and decode
    prompt / completion pairs (the reference prints the first five as a table and returns nothing; here they are returned).  Sampling goes
    through the trainer's own `actor_step` -- the native `generate` with the rollout's settings -- where the reference calls
    `generate(max_length=model_max_length, do_sample=True)` with the model's default generation config.  {} without an evaluation dataloader."""
    dl = eval_dataloader if eval_dataloader is not None else getattr(trainer, 'eval_dataloader', None)
    if dl is None:
        return {}
    tok = getattr(trainer, 'tokenizer', None)
    prompts, generateds = [], []
    for batch in dl:
        out = trainer.actor_step(batch)
        seq = (out[0] if isinstance(out, tuple) else out)['input_ids']
        if tok is None:                                  # no tokenizer (model_cfg + state constructors): token ids
            P = batch['input_ids'].shape[1]
            prompts.extend(batch['input_ids'].tolist())
            generateds.extend(seq[:, P:].tolist() if seq.shape[1] >= P else seq.tolist())
            continue
        prompt = tok.batch_decode(batch['input_ids'], skip_special_tokens=True)
        text = tok.batch_decode(seq, skip_special_tokens=True)
        prompts.extend(prompt)
        generateds.extend(t[len(prompt[i]):] for i, t in enumerate(text))
    return {'eval/prompts': prompts, 'eval/generated': generateds}


def resolve_pretrained(cfgs, device, *, trainable, head='lm', dtype=None, path_key='model_cfgs.model_name_or_path', build_kwargs=None,
                       with_tokenizer=True):
    """The reference trainers load their own models from `model_cfgs.model_name_or_path` (text_to_text/dpo.py:83-100,
    text_image_to_text/dpo.py:58-83 -> models/pretrained_model.py:160-312); the native ones do the same through checkpoint.load_pretrained
    (config.json -> native geometry, sharded safetensors / .bin streamed into the flat device buffers, tokenizer / processor, pad-token
    resize).  Returns (model, tokenizer, processor, hf_config)."""
    from ..checkpoint import load_pretrained
    path = cfg_get(cfgs, path_key, None)
    if not path:
        raise ValueError(f'{path_key} is not set and no model_cfg / state was handed to the trainer')
    return load_pretrained(path, device, trainable=trainable, head=head, dtype=dtype or compute_dtype(cfg_get(cfgs, 'train_cfgs.compute_dtype', 'bf16')),
                           model_max_length=int(cfg_get(cfgs, 'model_cfgs.model_max_length', 512)), padding_side='left',
                           build_kwargs=build_kwargs, with_tokenizer=with_tokenizer, processor_kwargs=cfg_get(cfgs, 'train_cfgs.processor_kwargs', None))


def infer_modality(trainer) -> str:
    """Which `align_anything.datasets.<modality>` package a trainer's batches come from.  The reference fixes it per trainer MODULE
    (trainers/text_image_to_text/dpo.py imports datasets.text_image_to_text, trainers/text_audio_to_text/dpo.py datasets.text_audio_to_text, ...);
    the native trainers are one class per algorithm, so: an explicit `trainer.modality` or `data_cfgs.modality` wins, else the policy's backbone
    decides -- Qwen2-Audio -> text_audio_to_text, any other model loaded with a processor -> text_image_to_text, else text_to_text."""
    explicit = getattr(trainer, 'modality', None) or cfg_get(trainer.cfgs, 'data_cfgs.modality', None)
    if explicit:
        return str(explicit)
    kind = (getattr(trainer, 'model_cfg', None) or {}).get('kind')
    if kind is None:
        for name in ('actor_model', 'model'):
            kind = getattr(getattr(getattr(trainer, name, None), 'module', None), 'kind', None) or kind
    if kind == 'qwen2audio':
        return 'text_audio_to_text'
    return 'text_image_to_text' if getattr(trainer, 'processor', None) is not None else 'text_to_text'


def get_dataloaders(trainer, train_dtype_name: str, eval_dtype_name: str | None = None, modality: str | None = None, ptx_dtype_name: str | None = None,
                    rl: bool = False):
    """`SupervisedTrainerBase.get_dataloaders` (base/supervised_trainer.py:79-232) and, with rl=True, `RLTrainerBase.get_dataloaders`
    (base/rl_trainer.py:78-172) for the native trainers' `init_datasets()`.

    The dataset / template plugin surface stays the reference's own (`@register_template` formatters, the Dataset
    constructor signature, `get_collator()`): the classes are imported from the installed `align_anything` package and called exactly as the
    reference does -- ChatTemplate(formatter, template), Dataset(path, template, tokenizer, processor, name, size, split, data_files,
    optional_args), DataLoader(collate_fn=dataset.get_collator(), sampler=DistributedSampler(shuffle=True), batch_size=...) -- and the loader is
    wrapped in the device prefetcher (data.DevicePrefetcher: next batch host -> HBM on a side stream, window plan prebuilt).  Batch sizes as in the
    reference: supervised train / eval = per_device_train / eval_batch_size; RL prompts = per_device_prompt_batch_size, RL eval and PTX =
    per_device_train_batch_size.  `data_cfgs.*_datasets` as a LIST (several datasets, one template each) becomes a ConcatDataset with the first
    dataset's collator (supervised_trainer.py:110-160).  Single-process runs use a DistributedSampler of one replica (the order a world-1
    reference run sees).  Returns (train, eval) loaders, or (train, eval, ptx) with ptx_dtype_name; None where data_cfgs names no dataset."""
    import importlib
    import torch.distributed as dist
    from torch.utils.data import ConcatDataset, DataLoader
    from torch.utils.data.distributed import DistributedSampler
    from ..data import DevicePrefetcher
    cfgs = trainer.cfgs
    d = lambda k, default=None: cfg_get(cfgs, 'data_cfgs.' + k, default)
    want = ('train', 'eval') + (('ptx',) if ptx_dtype_name else ())
    if not any(d(p + '_datasets') for p in want):
        return (None,) * len(want)
    if modality is None:
        modality = infer_modality(trainer)
    try:
        ds_mod = importlib.import_module(f'align_anything.datasets.{modality}')
        from align_anything.configs.template import ChatTemplate
    except ImportError as e:
        raise RuntimeError('init_datasets(): data_cfgs names datasets, which are built by the reference\'s own dataset / template plugins '
                           f'(align_anything.datasets.{modality}, align_anything.configs.template) -- that package is not importable here ({e}); '
                           'install it, or hand the trainer dataloaders of collated batches') from e
    tokenizer, processor = getattr(trainer, 'tokenizer', None), getattr(trainer, 'processor', None)
    formatter = processor if processor else tokenizer
    custom = getattr(getattr(trainer, 'hf_model_hooks', None), 'apply_chat_template', None)
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    pick = lambda v, i: (v[i] if isinstance(v, (list, tuple)) and v else (None if isinstance(v, (list, tuple)) else v))

    def build(prefix, dtype_name, batch_key):
        paths = d(prefix + '_datasets')
        if not paths:
            return None
        cls = getattr(ds_mod, dtype_name)
        many = not isinstance(paths, str)
        templates, sets = [], []
        for i, path in enumerate(list(paths) if many else [paths]):
            tname = pick(d(prefix + '_template'), i) if many else d(prefix + '_template')
            template = ChatTemplate(formatter, tname, None if many else custom)
            templates.append(template)
            f = (lambda k: pick(d(prefix + '_' + k), i)) if many else (lambda k: d(prefix + '_' + k))
            sets.append(cls(path=path, template=template, tokenizer=tokenizer, processor=processor, name=f('name'), size=f('size'), split=f('split'),
                            data_files=f('data_files'), optional_args=(f('optional_args') or []) if many else d(prefix + '_optional_args', [])))
        setattr(trainer, prefix + '_template', templates if many else templates[0])
        ds = ConcatDataset(sets) if many else sets[0]
        loader = DataLoader(ds, collate_fn=sets[0].get_collator(), sampler=DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True),
                            batch_size=int(cfg_get(cfgs, 'train_cfgs.' + batch_key, 1)))
        return DevicePrefetcher(loader, trainer.device, getattr(trainer, 'pad_token_id', None))

    out = [build('train', train_dtype_name, 'per_device_prompt_batch_size' if rl else 'per_device_train_batch_size'),
           build('eval', eval_dtype_name or train_dtype_name, 'per_device_train_batch_size' if rl else 'per_device_eval_batch_size')]
    if ptx_dtype_name:
        out.append(build('ptx', ptx_dtype_name, 'per_device_train_batch_size'))
    return tuple(out)
