"""Native autoregressive rollout: the replacement of HF `generate` in the PPO loop
(align_anything/trainers/text_to_text/ppo.py:209-222, text_image_to_text/ppo.py:174-204).

Prefill = the ordinary native forward with a KV sink; every later token is one pass of HBM-streaming kernels
(csrc/decode.hip): skinny GEMMs, cache attention, norm, sampling.  HF semantics that are reproduced:
  * left-padded prompts; position ids derived from the attention mask (cumsum - 1), i.e. NOT the arange
    positions of the training forward (a reference quirk, SURVEY.md §8 a');
  * RepetitionPenaltyLogitsProcessor, then TemperatureLogitsWarper / TopPLogitsWarper;
  * finished rows keep emitting pad_token_id; generation stops when every row has produced EOS or the length
    cap (GenerationConfig.max_length = model_max_length) is reached; `synced_gpus` is not needed (pure DP).
"""
from __future__ import annotations

import torch

from . import ops


_HF_TOP_K = []


def hf_default_top_k():
    """`GenerationConfig().top_k` of the installed transformers (cached): the value the reference's rollouts sample with when its yaml does
    not set one."""
    if not _HF_TOP_K:
        try:
            from transformers import GenerationConfig
            _HF_TOP_K.append(GenerationConfig().top_k)
        except Exception:
            _HF_TOP_K.append(None)
    return _HF_TOP_K[0]


@torch.no_grad()
def generate(model, input_ids, attention_mask, *, max_length=None, max_new_tokens=None, do_sample=True,
             temperature=1.0, top_p=1.0, repetition_penalty=1.0, eos_token_id=None, pad_token_id=0,
             pixel_values=None, generator=None, sync_every=8, use_graph=False, top_k='hf', **mm):
    """Returns sequences [N, T_prompt + n_new] (int64), right-padded with pad_token_id after EOS.
    top_k: HF's TopKLogitsWarper between temperature and top-p; 'hf' = whatever the installed transformers' GenerationConfig defaults to
    (the reference builds GenerationConfig(temperature, top_p, repetition_penalty, do_sample=True) and inherits it: 50 under 4.x, None under
    5.x -- trainers/text_to_text/ppo.py:161-170); None / 0 = no cut."""
    if top_k == 'hf':
        top_k = hf_default_top_k()
    if repetition_penalty <= 0.0:
        raise ValueError('repetition_penalty must be > 0')
    N, T = input_ids.shape
    dev = input_ids.device
    if getattr(model, 'dtype', torch.bfloat16) != torch.bfloat16:
        raise RuntimeError('generate: the HIP decode kernels are bf16 only (the fp32 parity mode covers training steps)')
    if max_new_tokens is None:
        if max_length is None:
            raise ValueError('give max_length (GenerationConfig.max_length) or max_new_tokens')
        max_new_tokens = max_length - T
    stack = model.stack
    # Expert-parallel weights (expert_parallel.py): every decode position is a token exchange between ALL ranks, so the ranks must
    # execute the same number of passes whatever their prompts and EOS positions are -- what `synced_gpus=True` does for the
    # reference's ZeRO-3 rollouts (text_to_text/ppo.py:209-222).  The pass count is the maximum over the ranks; a rank whose own
    # budget (max_length - its prompt length) is used up, or whose rows have all finished, keeps stepping on pad tokens, and the
    # loop ends early only when NO rank has an unfinished row.
    ep = getattr(stack, 'ep', None)
    own_new = max_new_tokens
    if ep is not None and ep.size > 1:
        import torch.distributed as dist
        # one exchange decides for ALL ranks: [max, -min] of the ranks' own budgets.  A rank whose prompts already fill the length cap
        # would return its prompts unchanged (zero generated columns) and fail in the caller's window plan while the other ranks sit
        # in the token exchange of their first decode position -- so everybody raises, together, before any exchange
        cnt = torch.tensor([max(max_new_tokens, 0), -max_new_tokens], dtype=torch.int64, device=dev if not ep.host_staged else 'cpu')
        dist.all_reduce(cnt, op=dist.ReduceOp.MAX, group=ep.group)
        max_new_tokens, min_new = int(cnt[0].item()), -int(cnt[1].item())
        if min_new <= 0:
            raise ValueError(f'generate (expert-parallel): a rank has no room for new tokens (smallest budget {min_new}, this rank {own_new}): '
                             'prompt length >= max_length on that rank; every rank raises so that nobody waits in a collective')
    else:
        ep = None
    if max_new_tokens <= 0:
        return input_ids
    Tmax = T + max_new_tokens
    kvw = stack.kv_width()
    cache = [torch.zeros((N * Tmax, kvw), dtype=torch.bfloat16, device=dev) for _ in stack.layers]

    am = attention_mask.to(torch.int64)
    valid = am.sum(dim=1)                                    # real prompt tokens per row
    position_ids = (torch.cumsum(am, dim=1) - 1).clamp_(min=0)
    start = am.to(torch.int32).argmax(dim=1).to(torch.int32)

    def kv_sink(li, kv):  # kv: [N*T, kvw] of layer li
        cache[li].view(N, Tmax, kvw)[:, :T] = kv.view(N, T, kvw)

    # **mm: extra processor outputs of the backbone (Qwen2-VL: image_grid_thw; its prompt positions are the 3-D rope index,
    # generated tokens continue at max position + 1 on all three axes = 1-D RoPE, hf:models/qwen2_vl/modeling_qwen2_vl.py:1122-1136)
    x = model.forward_stream(input_ids, attention_mask, pixel_values, save=False, position_ids=position_ids,
                             kv_sink=kv_sink, **mm)
    last_rows = torch.arange(N, device=dev) * T + (T - 1)
    # the weights are frozen for the whole rollout: stacks that support it hand the strip kernel copies in its own order
    dw = stack.prepare_decode(N) if hasattr(stack, 'prepare_decode') else None
    step_kw = {} if dw is None else {'dw': dw}
    head_w = model.head.prepare_decode(N) if (dw is not None and hasattr(model.head, 'prepare_decode')) else None
    logits = model.head.logits_rows(ops.embed_fwd(last_rows, x), head_w) if head_w is not None else model.head.logits_rows(ops.embed_fwd(last_rows, x))

    out = torch.full((N, Tmax), pad_token_id, dtype=torch.int64, device=dev)
    out[:, :T] = input_ids
    is_opt = getattr(model, 'kind', '') == 'opt'
    eos = -1 if eos_token_id is None else int(eos_token_id)
    # ---- static device state of the decode loop (everything a step needs lives in these buffers, so one step
    # is a fixed launch sequence: captured once in a hipGraph and replayed -- the per-step host cost of ~300
    # ctypes/torch launches (launch-bound at 7B: ~17 us x 320) collapses to one graph launch)
    st = {
        'logits': logits.clone(), 'unfinished': torch.ones(N, dtype=torch.bool, device=dev),
        'tslot': torch.full((N,), T, dtype=torch.int64, device=dev), 'pos': model.decode_start_positions(valid).to(torch.int32).clone(),
        'length': torch.full((N,), T + 1, dtype=torch.int32, device=dev), 'step': torch.zeros(1, dtype=torch.int64, device=dev),
        'U': torch.rand((max_new_tokens, N), device=dev, generator=generator) if do_sample else None,
        # number of steps that still produced a token for at least one row = the columns HF's stopping criteria keep
        'nact': torch.zeros(1, dtype=torch.int64, device=dev),
    }
    padv = torch.full((N,), pad_token_id, dtype=torch.int64, device=dev)
    seen = None
    if repetition_penalty != 1.0:   # HF penalises every id of `input_ids` (left-pad ids included) and every generated one
        seen = ops.mark_seen_(torch.zeros((N, logits.shape[1]), dtype=torch.uint8, device=dev), input_ids.contiguous())

    def select(lg, u):
        if do_sample:
            return ops.sample_top_p(lg, temperature, top_p, u, seen, repetition_penalty, top_k=top_k)
        return ops.argmax_rows(lg, seen, repetition_penalty)

    def one_step():
        """select token from st['logits'], record it, run one decode pass, leave next logits in st['logits']."""
        nxt = select(st['logits'], st['U'].index_select(0, st['step'])[0] if do_sample else None)
        # nact += any(unfinished); nxt = where(unfinished, nxt, pad); out[n, tslot[n]] = nxt[n]; unfinished &= nxt != eos -- one launch (aa_decode_record)
        nxt = ops.decode_record(nxt, st['unfinished'], out, st['tslot'], st['nact'], pad_token_id, eos)
        if seen is not None:
            ops.mark_seen_(seen, nxt[:, None])
        emb_pos = (st['pos'] + 2) if is_opt else None        # OPT learned positions carry an offset of 2
        xt = model.embed_tokens(nxt, emb_pos)
        xt = stack.decode_step(xt, cache, st['tslot'], Tmax, st['pos'], start, st['length'], **step_kw)
        if head_w is not None:
            model.head.logits_rows(xt, head_w, out=st['logits'])       # the strip kernel writes the next position's logits in place
        else:
            st['logits'].copy_(model.head.logits_rows(xt))
        ops.decode_tick(st['tslot'], st['pos'], st['length'], st['step'])

    graph = None
    if use_graph and max_new_tokens > 4 and ep is None:      # collectives cannot be captured: expert-parallel rollouts launch eagerly
        # everything the warm-up step mutates is restored afterwards: the decode state, the output and the repetition-penalty
        # marks (`seen`), so a token the warm-up happened to sample is not penalised for the whole rollout
        snap = {k: v.clone() for k, v in st.items() if v is not None and k != 'U'}
        out_snap = out.clone()
        seen_snap = seen.clone() if seen is not None else None

        def restore():
            for k, v in snap.items():
                st[k].copy_(v)
            out.copy_(out_snap)
            if seen is not None:
                seen.copy_(seen_snap)

        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                one_step()                                      # warm-up on a side stream (allocator, lazy inits)
            torch.cuda.current_stream().wait_stream(side)
            restore()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                one_step()
            restore()                                           # capture does not execute: state is still pristine, but make that explicit
        except Exception:                                       # capture unsupported -> eager launches
            graph = None
            restore()

    generate.last_used_graph = graph is not None
    n_new = 0
    # the last selected token needs no decode pass, but running it keeps the loop a single replayed graph;
    # the cache has Tmax slots, so the final pass writes slot Tmax-1 at most
    # expert-parallel decode positions all exchange N x k pairs: the ranks (whose N may differ) agree on the block size once per rollout
    import contextlib
    scope = ep.pass_scope(N * stack.cfg['num_experts_per_tok']) if (ep is not None and ep.padded) else contextlib.nullcontext()
    with scope:
        for step in range(max_new_tokens):
            if step + 1 == max_new_tokens:
                # final token: selection only
                if ep is not None and step >= own_new:
                    st['unfinished'].zero_()
                st['nact'].add_(st['unfinished'].any().to(torch.int64))
                nxt = select(st['logits'], st['U'][step] if do_sample else None)
                nxt = torch.where(st['unfinished'], nxt, padv)
                out[:, T + step] = nxt
                n_new = step + 1
                break
            if ep is not None and step >= own_new:
                st['unfinished'].zero_()                            # this rank's own length cap: its rows only pad from here on
            if graph is not None:
                graph.replay()
            else:
                one_step()
            n_new = step + 1
            if eos >= 0 and (step % sync_every == sync_every - 1):
                if ep is None:
                    if not bool(st['unfinished'].any()):
                        break
                else:                                               # stop only when every rank is done (same decision on every rank)
                    flag = st['unfinished'].any().to(torch.int32).reshape(1)
                    flag = flag.cpu() if ep.host_staged else flag
                    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=ep.group)
                    if int(flag.item()) == 0:
                        break
    n_new = min(n_new, max(own_new, 0))
    seq = out[:, :T + n_new]
    if eos_token_id is not None:
        # HF stops at the step in which the last unfinished row emitted its EOS: keep exactly the columns of steps that still had
        # an unfinished row when they started (steps run past that point between two sync checks only wrote pad).  Counting steps,
        # not non-pad columns, keeps a final EOS column when pad_token_id == eos_token_id.
        seq = seq[:, :T + min(n_new, int(st['nact'].item()))]
    return seq
