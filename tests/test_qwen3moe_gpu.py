"""GPU: native Qwen3-MoE (BASELINE configs[4] backbone) against the fixture the reference's text_to_text DPOTrainer produced on HF
Qwen3MoeForCausalLM (tests/golden/qwen3moe_tiny_dpo.npz: 8 experts, top-2, q/k head RMSNorm, GQA): routing bit-exact, logits,
loss and every gradient incl. router and per-expert weights."""
import numpy as np
import pytest
import torch

from tests.gpu_util import assert_close, dev, dump
from tests.util import load_golden, rel_err, state_dict_from_golden, tiny_qwen3moe_cfg

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def test_moe_route_tie_rule_is_pinned():
    """the router's top-k TIE behaviour, stated and pinned.  bf16 router logits collide often (8 mantissa bits, 128 experts), and
    equal logits are equal probabilities; hf's `torch.topk(routing_weights, top_k)` (modeling_qwen3_moe.py:210-282, what models/qwen3_moe.py:28-60 runs)
    has NO defined order among equal values -- on CPU it differs from index-ascending in ~20 % of rows with a tie at the k-th place (measured in the build
    container), so no fixture can pin it.  aa_moe_route's rule is: probability descending, then expert index ASCENDING (csrc/moe.hip: k rounds of a
    wave-wide argmax with the (value, -index) order) -- deterministic, identical on every rank of an expert-parallel job, and a valid top-k wherever the
    reference's is: the selected VALUES are the same multiset.  Rows without ties equal torch.topk exactly (test_moe_kernels_vs_torch)."""
    from align_anything_amd import ops
    g = torch.Generator().manual_seed(11)
    rows, E, k = 512, 128, 8
    logits = torch.randn(rows, E, generator=g).to(torch.bfloat16)
    for r in range(rows):                                   # force ties, in most rows AT the k-th place: copy the k-th largest logit into 1 - 6 other experts
        v, i = torch.topk(logits[r].float(), k)
        n = int(torch.randint(1, 7, (1,), generator=g))
        others = torch.randperm(E, generator=g)[:n]
        logits[r, others] = (logits[r, i[k - 1]] if r % 4 else logits[r, i[0]]).clone()
    probs, idx, w = ops.moe_route(logits.to(dev()), k, True)
    p = torch.softmax(logits.float(), -1)
    order = torch.tensor([sorted(range(E), key=lambda e: (-float(p[r, e]), e))[:k] for r in range(rows)])
    assert torch.equal(idx.long().cpu(), order), 'tie rule: probability descending, then expert index ascending'
    tv, ti = torch.topk(p, k, -1)
    assert torch.equal(torch.sort(torch.gather(p, 1, order), -1).values, torch.sort(tv, -1).values)       # the same VALUES as the reference's call, whatever it picked
    n_boundary = sum(float(p[r, order[r, -1]]) == float(torch.sort(p[r], descending=True).values[k]) for r in range(rows))
    assert n_boundary > rows // 2                            # the case is exercised: a tie across the k-th place in most rows
    sel = torch.gather(p, 1, order)
    assert rel_err(w.float().cpu(), (sel / sel.sum(-1, keepdim=True))) < 1e-2
    again = ops.moe_route(logits.to(dev()), k, True)[1]
    assert torch.equal(again, idx)


def test_moe_kernels_vs_torch():
    from align_anything_amd import ops
    g = torch.Generator().manual_seed(2)
    rows, E, k, h = 100, 16, 4, 64
    logits = (torch.randn(rows, E, generator=g) * 2).to(dev())
    for norm in (True, False):
        probs, idx, w = ops.moe_route(logits, k, norm)
        p_ref = torch.softmax(logits.double(), -1)
        tv, ti = torch.topk(p_ref, k, -1)
        assert torch.equal(idx.long().cpu(), ti.cpu())                                   # integer work: exact
        wr = tv / tv.sum(-1, keepdim=True) if norm else tv
        assert rel_err(w.cpu(), wr.cpu()) < 1e-6 and rel_err(probs.cpu(), p_ref.cpu()) < 1e-6
        # backward of the routing weights w.r.t. the logits
        lg = logits.double().clone().requires_grad_(True)
        pr = torch.softmax(lg, -1)
        sel = torch.gather(pr, 1, ti)
        ww = sel / sel.sum(-1, keepdim=True) if norm else sel
        dw = torch.randn(rows, k, generator=g).to(dev())
        (ww * dw.double()).sum().backward()
        got = ops.moe_route_bwd(probs, idx, dw, norm, torch.float32)
        assert rel_err(got.cpu(), lg.grad.cpu()) < 1e-5
    # device-side plan (no host read) vs the same layout computed with torch: integer work, exact
    plan = ops.moe_plan(idx, E)
    A = ops.MOE_ALIGN
    flat = idx.reshape(-1).long().cpu()
    counts = torch.bincount(flat, minlength=E)
    seg = (counts + A - 1) // A * A
    off = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(seg, 0)])
    assert plan['counts'].cpu().tolist() == counts.tolist() and plan['off'].cpu().tolist() == off.tolist()
    order = torch.argsort(flat, stable=True)
    starts = torch.cumsum(counts, 0) - counts
    dest = off[flat[order]] + (torch.arange(rows * k) - starts[flat[order]])
    want_pos = torch.empty(rows * k, dtype=torch.long); want_pos[order] = dest
    assert torch.equal(plan['pos'].long().cpu().reshape(-1), want_pos)
    want_src = torch.full((plan['cap'],), -1, dtype=torch.long); want_src[dest] = order // k
    assert torch.equal(plan['src'].long().cpu(), want_src)
    TG = ops.MOE_TILE_ROWS                       # the tile table has one entry per 128 rows, whatever the segment alignment
    te = torch.full((plan['cap'] // TG,), -1, dtype=torch.long)
    for e in range(E):
        te[off[e] // TG: off[e + 1] // TG] = e
    assert torch.equal(plan['tile_expert'].long().cpu(), te)
    x = torch.randn(rows, h, generator=g).to(dev())
    xp = ops.moe_gather(x, plan['src'])
    src = plan['src'].cpu()
    assert torch.equal(xp.cpu()[src >= 0], x.cpu()[src[src >= 0].long()]) and float(xp.cpu()[src < 0].abs().max()) == 0.0
    pos = plan['pos'].long().cpu()
    # grouped GEMMs (all experts in one launch) vs a per-expert loop in torch
    for dt, tol in ((torch.float32, 1e-5), (torch.bfloat16, 2e-2)):
        F_ = 128
        w3 = (torch.randn(E, F_, h, generator=g) * 0.2).to(dt).to(dev())
        xq = xp.to(dt)
        out = ops.gemm_grouped(xq, w3, plan, out=torch.zeros((plan['cap'], F_), dtype=dt, device=dev()))
        ref = torch.zeros(plan['cap'], F_, dtype=torch.float64)
        for e in range(E):
            ref[off[e]:off[e + 1]] = xq[off[e]:off[e + 1]].double().cpu() @ w3[e].double().cpu().t()
        assert rel_err(out.cpu()[: off[E]], ref[: off[E]]) < tol
        dy = (torch.randn(plan['cap'], F_, generator=g) * (src >= 0)[:, None]).to(dt).to(dev())
        dx = ops.gemm_grouped(dy, w3, plan, out=torch.zeros((plan['cap'], h), dtype=dt, device=dev()), b_n=True)
        refx = torch.zeros(plan['cap'], h, dtype=torch.float64)
        for e in range(E):
            refx[off[e]:off[e + 1]] = dy[off[e]:off[e + 1]].double().cpu() @ w3[e].double().cpu()
        assert rel_err(dx.cpu()[: off[E]], refx[: off[E]]) < tol
        gw = torch.full((E, F_, h), 7.0, dtype=dt, device=dev())                    # stale values must be overwritten, also for empty experts
        ops.gemm_grouped_dw(dy, xq, plan, gw)
        refw = torch.stack([dy[off[e]:off[e + 1]].double().cpu().t() @ xq[off[e]:off[e + 1]].double().cpu() for e in range(E)])
        assert rel_err(gw.cpu(), refw) < tol
        for e in range(E):
            if counts[e] == 0:
                assert float(gw[e].float().abs().max()) == 0.0
    yp = torch.randn(xp.shape[0], h, generator=g).to(dev())
    res = torch.randn(rows, h, generator=g).to(dev())
    out = ops.moe_combine(yp, plan['pos'], w, rows, residual=res)
    ref = res.cpu().double() + (yp.cpu().double()[pos] * w.cpu().double()[..., None]).sum(1)
    assert rel_err(out.cpu(), ref) < 1e-6
    dout = torch.randn(rows, h, generator=g).to(dev())
    dyp, dw = ops.moe_combine_bwd(dout, yp, plan['pos'], w)
    assert rel_err(dw.cpu(), (dout.cpu().double()[:, None] * yp.cpu().double()[pos]).sum(-1)) < 1e-6
    want = torch.zeros_like(yp.cpu().double())
    want[pos.reshape(-1)] = (dout.cpu().double()[:, None] * w.cpu().double()[..., None]).reshape(rows * k, h)
    assert rel_err(dyp.cpu(), want) < 1e-6


def test_moe_plan_and_combine_at_scale():
    """Round-3 kernels: the plan scans 4096 pairs per trip (16 per thread, one barrier) -- full trips, the ragged last trip, pairs without an expert
    (-1, the zero tail of the capacity-padded exchange) and empty experts, against the stable sort that defines it; the combine requests all k rows
    before it adds them in ascending-row order -- bit-exact against the rounding sequence written out in torch, incl. slots without a row."""
    from align_anything_amd import ops
    g = torch.Generator().manual_seed(4)
    for rows, E, k, invalid in ((3000, 128, 8, False), (1029, 24, 2, True), (4096, 8, 1, True), (512, 128, 8, False)):
        idx = torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(rows)]).to(torch.int32)
        idx[idx == 5] = 6 if E > 8 else 5                                      # expert 5 empty (E > 8)
        if invalid:
            idx[torch.rand(rows, k, generator=g) < 0.3] = -1
        plan = ops.moe_plan(idx.to(dev()), E, allow_invalid=invalid)
        A = ops.MOE_ALIGN
        flat = idx.reshape(-1).long()
        valid = flat >= 0
        counts = torch.bincount(flat[valid], minlength=E)
        seg = (counts + A - 1) // A * A
        off = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(seg, 0)])
        assert plan['counts'].cpu().tolist() == counts.tolist() and plan['off'].cpu().tolist() == off.tolist()
        order = torch.argsort(torch.where(valid, flat, E), stable=True)[:int(valid.sum())]
        starts = torch.cumsum(counts, 0) - counts
        dest = off[flat[order]] + (torch.arange(order.numel()) - starts[flat[order]])
        want_pos = torch.full((rows * k,), -1, dtype=torch.long); want_pos[order] = dest
        assert torch.equal(plan['pos'].long().cpu().reshape(-1), want_pos), (rows, E, k)
        want_src = torch.full((plan['cap'],), -1, dtype=torch.long); want_src[dest] = order // k
        assert torch.equal(plan['src'].long().cpu(), want_src), (rows, E, k)
        # bf16 combine: acc = bf16(acc + bf16(y w)) over the slots in ascending row order, then bf16(acc + residual)
        h = 64
        yp = torch.randn(plan['cap'], h, generator=g).to(torch.bfloat16)
        w = torch.rand(rows, k, generator=g).to(torch.bfloat16)
        res = torch.randn(rows, h, generator=g).to(torch.bfloat16)
        pos = want_pos.view(rows, k)
        key = torch.where(pos >= 0, pos, torch.full_like(pos, 1 << 40))
        srt = torch.argsort(key, dim=1)
        acc = torch.zeros(rows, h)
        rb = lambda t: t.to(torch.bfloat16).float()
        for jj in range(k):
            j = srt[:, jj]
            pj = pos[torch.arange(rows), j]
            term = rb(yp.float()[pj.clamp(min=0)] * w.float()[torch.arange(rows), j][:, None])
            acc = torch.where((pj >= 0)[:, None], rb(acc + term), acc)
        for use_w, use_res in ((True, True), (False, False)):
            got = ops.moe_combine(yp.to(dev()), plan['pos'], w.to(dev()) if use_w else None, rows, residual=res.to(dev()) if use_res else None)
            if use_w:
                want = (acc + res.float()).to(torch.bfloat16)
                assert torch.equal(got.cpu().view(torch.int16), want.view(torch.int16)), (rows, E, k, float((got.cpu().float() - want.float()).abs().max()))
            elif k == 1:
                want = torch.where((pos >= 0), yp.float()[pos.clamp(min=0).view(-1)], torch.zeros(rows, h)).to(torch.bfloat16)
                assert torch.equal(got.cpu().view(torch.int16), want.view(torch.int16))
        # grouped GEMM without a memset: tiles that belong to no expert come out as zeros, never as stale memory
        F_ = 256
        w3 = (torch.randn(E, F_, h, generator=g) * 0.2).to(torch.bfloat16).to(dev())
        xp = ops.moe_gather(torch.randn(rows, h, generator=g).to(torch.bfloat16).to(dev()), plan['src'])
        stale = torch.full((plan['cap'], F_), 7.0, dtype=torch.bfloat16, device=dev())
        out = ops.gemm_grouped(xp, w3, plan, out=stale)
        assert float(out[int(off[E]):].float().abs().max()) == 0.0 if int(off[E]) < plan['cap'] else True
        srcs = plan['src'].cpu()
        assert float(out.cpu()[srcs < 0].float().abs().max()) == 0.0


def _trainer(z, dtype):
    from align_anything_amd.trainers.dpo import DPOTrainer
    cfgs = {'train_cfgs': {'scale_coeff': float(z['scale_coeff']), 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant',
                           'weight_decay': 0.0, 'compute_dtype': dtype},
            'model_cfgs': {'pad_token_id': int(z['pad_token_id'])}}
    wd = torch.bfloat16 if dtype == 'bf16' else torch.float32
    return DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=tiny_qwen3moe_cfg(), policy_state=state_dict_from_golden(z, 'w.', wd),
                      reference_state=state_dict_from_golden(z, 'r.', wd), device='cuda:0')


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_qwen3moe_dpo_matches_reference_fixture(dtype):
    z = load_golden('qwen3moe_tiny_dpo.npz')
    tr = _trainer(z, dtype)
    tight = dtype == 'fp32'
    b = {'input_ids': T(z['input_ids']).to(dev()), 'attention_mask': T(z['attention_mask']).to(dev()),
         'meta_info': {'response_lens': [int(x) for x in z['response_lens']]}}
    logits = tr.policy.logits(b['input_ids'], b['attention_mask']).float().cpu()
    valid = T(z['attention_mask']).bool()
    e_log = rel_err(logits[valid], T(z['policy_logits'])[valid])
    rep = [f'{dtype}: logits rel_err {e_log:.2e}']
    assert e_log < (2e-5 if tight else 4e-2), rep
    lp = tr.compute_log_probs(tr.model, b).cpu()
    assert torch.equal(lp == 0, T(z['seq_log_probs']) == 0)
    e_lp = float((lp - T(z['seq_log_probs'])).abs().max())
    rep.append(f'max |d log-prob| per token {e_lp:.2e}')
    assert e_lp < (1e-4 if tight else 4.5e-2), rep      # bf16 measured 2.2e-2
    ld = tr.loss(b)
    rep.append(f"loss native {float(ld['loss']):.6f} reference {float(z['loss_loss']):.6f}")
    assert abs(float(ld['loss']) - float(z['loss_loss'])) < (3e-5 if tight else 2e-2)
    tr.model.backward(ld['loss'])
    torch.cuda.synchronize()
    worst, n = 0.0, 0
    for k in z.files:
        if not k.startswith('g.'):
            continue
        g = tr.policy.store.grad_view(k[2:])
        assert g is not None, k
        want = T(z[k])
        got = g.float().cpu().reshape(want.shape)
        if float(want.norm()) < 1e-6:
            assert float(got.norm()) < 1e-4, k
            continue
        e = rel_err(got, want)
        worst = max(worst, e); n += 1
        assert e < (5e-4 if tight else 1.2e-1), (k, e)
    rep.append(f'worst gradient rel_err {worst:.2e} over {n} tensors (router, experts, q/k norms, attention, embeddings)')
    dump(f'parity_qwen3moe_{dtype}.txt', '\n'.join(rep) + '\n')
    assert n >= 25
    info = tr.train_step(b)
    assert np.isfinite(info['train/loss'])


def test_grpo_step_on_qwen3moe_backbone():
    """BASELINE configs[4] names GRPO on Qwen3-MoE: the native GRPOTrainer (trainers/grpo.py, pinned to the reference's own
    train_step on OPT) runs unchanged on the MoE backbone -- loss against the oracle on the same sequences / rewards."""
    from oracle import models as om
    from oracle import rl_math as orl
    from align_anything_amd.trainers.grpo import GRPOTrainer
    z = load_golden('qwen3moe_tiny_dpo.npz')
    cfg = tiny_qwen3moe_cfg()
    sd, rsd = state_dict_from_golden(z, 'w.'), state_dict_from_golden(z, 'r.')
    cfgs = {'train_cfgs': {'actor_lr': 1e-3, 'actor_weight_decay': 0.0, 'actor_lr_warmup_ratio': 0.0, 'actor_lr_scheduler_type': 'constant',
                           'beta': 0.04, 'num_generations': 3, 'compute_dtype': 'fp32'},
            'model_cfgs': {'pad_token_id': 1, 'eos_token_id': 2}}
    tr = GRPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, actor_state=sd, reference_state=rsd, reward_fn=lambda c: c.sum(1).float(),
                     device='cuda:0')
    g = torch.Generator().manual_seed(3)
    B, G, P, L = 2, 3, 10, 12
    prompts = torch.randint(3, 320, (B, P), generator=g)
    seqs = torch.cat([prompts.repeat_interleave(G, 0), torch.randint(3, 320, (B * G, L), generator=g)], 1)
    seqs[1, P + 5] = 2; seqs[1, P + 6:] = 1
    rewards = torch.randn(B * G, generator=g)
    info = tr.train_step({'input_ids': prompts.to(dev()), 'attention_mask': torch.ones_like(prompts).to(dev())},
                         sequences=seqs.to(dev()), rewards=rewards.to(dev()))
    am = (seqs != 1).long()
    lp = orl.gather_log_probabilities(om.qwen3moe_logits(sd, cfg, seqs, am)[:, :-1][:, -L:], seqs[:, -L:])
    rlp = orl.gather_log_probabilities(om.qwen3moe_logits(rsd, cfg, seqs, am)[:, :-1][:, -L:], seqs[:, -L:])
    want, _, _ = orl.grpo_loss(lp, rlp, rewards, B, G, seqs[:, P:], 2, 0.04)
    assert abs(info['train/loss'] - float(want)) < 5e-5, (info['train/loss'], float(want))
    assert abs(info['train/reward'] - float(rewards.mean())) < 1e-6


def test_moe_gemv_rows_stream_their_own_expert():
    """aa_moe_gemv_bf16 (decode): out[r] = x[r // x_div] @ W3[row_expert[r]]^T."""
    from align_anything_amd import ops
    g = torch.Generator().manual_seed(5)
    for (E, N, K, R, xd) in [(8, 128, 128, 8, 2), (128, 1536, 2048, 32, 8), (16, 2048, 768, 24, 1), (4, 100, 64, 5, 1)]:
        w3 = (torch.randn(E, N, K, generator=g) * 0.1).to(torch.bfloat16).to(dev())
        x = torch.randn((R + xd - 1) // xd, K, generator=g).to(torch.bfloat16).to(dev())
        re = torch.randint(0, E, (R,), generator=g).to(torch.int32).to(dev())
        out = ops.moe_gemv(x, w3, re, xd)
        ref = torch.stack([x[r // xd].double().cpu() @ w3[int(re[r])].double().cpu().t() for r in range(R)])
        assert rel_err(out.float().cpu(), ref) < 6e-3, (E, N, K, R, xd)


def test_generate_greedy_qwen3moe_both_expert_paths():
    """Rollout on the MoE backbone (BASELINE configs[4] = GRPO on Qwen3-MoE): KV-cache prefill + one-token decode with the routed
    experts streamed per row (N*k small) and through the 128-row-tile layout (bigger batches), against the fp32 oracle's greedy
    continuation; then the decode path against the native full forward on the generated sequence (same bf16 arithmetic)."""
    from oracle import models as om
    from tests.test_decode_gpu import _check_greedy
    from align_anything_amd.modeling import build_model
    z = load_golden('qwen3moe_tiny_dpo.npz')
    cfg = tiny_qwen3moe_cfg()
    m = build_model(cfg, 'cuda:0', trainable=False)
    m.load_state_dict(state_dict_from_golden(z, 'w.', torch.bfloat16))
    sd = state_dict_from_golden(z, 'w.')
    fn = lambda i, a: om.qwen3moe_logits(sd, cfg, i, a)
    ids, mask = T(z['input_ids'])[:, :20].clone(), T(z['attention_mask'])[:, :20].clone()
    pad = int(z['pad_token_id'])
    seq = _check_greedy(m, fn, ids, mask, 8, None, pad, 'qwen3moe_gemv')
    big = _check_greedy(m, fn, ids.repeat(5, 1), mask.repeat(5, 1), 8, None, pad, 'qwen3moe_grouped')     # 20 rows -> tile layout
    assert big.shape[0] == 20 and seq.shape[1] == 28
    # decode vs the native training forward on the same tokens: logits of the last position of every prefix agree to bf16 noise
    full = m.logits(seq.to(dev()), torch.cat([mask, torch.ones(mask.shape[0], 8, dtype=mask.dtype)], 1).to(dev())).float().cpu()
    top2 = torch.topk(full[:, 19:-1], 2, dim=-1)
    sure = (top2.values[..., 0] - top2.values[..., 1]) > 0.1
    assert torch.equal(top2.indices[..., 0][sure], seq[:, 20:][sure]) and int(sure.sum()) >= 4


def test_capacity_padded_expert_parallel_block_never_reads_the_device():
    """the exact expert-parallel exchange reads the split sizes on the host once per MoE block and direction.  The
    capacity-padded form (expert_parallel.py, the trainers' default) must not: the whole decoder stack, forward and backward, runs with torch's
    sync-debug mode set to 'error' (any .item() / .cpu() / nonzero-style device->host read raises) on a communicator of one rank -- the exchange is then
    a copy, everything around it is the code every rank runs -- and the result is bit-identical to the stack without expert parallelism."""
    import torch.distributed as dist
    from align_anything_amd.expert_parallel import ExpertParallel
    from align_anything_amd.modeling import build_model
    z = load_golden('qwen3moe_tiny_dpo.npz')
    cfg = tiny_qwen3moe_cfg()
    own = not dist.is_initialized()
    if own:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29641', rank=0, world_size=1)
    try:
        ids, am = T(z['input_ids']).to(dev()), T(z['attention_mask']).to(dev())
        outs = {}
        for mode in ('plain', 'padded', 'exact'):
            ep = None if mode == 'plain' else ExpertParallel(dist.new_group(), capacity_factor=2.0 if mode == 'padded' else None, dense_below=0)
            m = build_model(cfg, 'cuda:0', trainable=True, dtype=torch.bfloat16, **({} if ep is None else {'ep': ep}))
            m.load_state_dict(state_dict_from_golden(z, 'w.', torch.bfloat16))
            m.store.init_training()
            N, Tn, Mp, start, pos = m._token_geometry(ids, am, None)
            flat = ids.reshape(-1)
            if Mp != N * Tn:
                flat = torch.cat([flat, torch.zeros(Mp - N * Tn, dtype=flat.dtype, device=flat.device)])
            x = m.embed_tokens(flat)
            dres = (torch.randn(x.shape, generator=torch.Generator().manual_seed(3)) * 0.05).to(torch.bfloat16).to(dev())
            m.stack.forward(x, N, Tn, start, pos, True)                # warm-up outside the guard: rotary tables and other one-time uploads
            m.stack.backward(dres.clone(), N, Tn, start, pos)
            m.store.zero_grad()
            torch.cuda.synchronize()
            if mode == 'padded':
                torch.cuda.set_sync_debug_mode('error')
            try:
                y = m.stack.forward(x, N, Tn, start, pos, True)
                dx = m.stack.backward(dres.clone(), N, Tn, start, pos)
                if ep is not None and ep.padded:
                    ep.poll_overflow()                           # the asynchronous flag read is not a sync either
            except RuntimeError as e:
                if mode == 'exact' and 'synchroniz' in str(e):
                    raise AssertionError('sync-debug mode was left on') from e
                raise
            finally:
                torch.cuda.set_sync_debug_mode('default')
            torch.cuda.synchronize()
            if ep is not None and ep.padded:
                ep.poll_overflow(block=True)
            outs[mode] = (y.clone(), dx.clone(), {n: m.store.grad_view(n).clone() for n in m.store.hf_names() if m.store.grad_view(n) is not None})
        # the exact exchange DOES read the device: the same guard trips on it (so the guard is live)
        ep = ExpertParallel(dist.new_group())
        m = build_model(cfg, 'cuda:0', trainable=False, dtype=torch.bfloat16, ep=ep)
        m.load_state_dict(state_dict_from_golden(z, 'w.', torch.bfloat16))
        torch.cuda.set_sync_debug_mode('error')
        try:
            with pytest.raises(RuntimeError):
                m.stack.forward(x, N, Tn, start, pos, False)
        finally:
            torch.cuda.set_sync_debug_mode('default')
        for mode in ('padded', 'exact'):
            assert torch.equal(outs[mode][0], outs['plain'][0]) and torch.equal(outs[mode][1], outs['plain'][1]), mode
            assert set(outs[mode][2]) == set(outs['plain'][2])
            for n, g in outs['plain'][2].items():
                got = outs[mode][2][n]
                # GEMM-made (bf16) gradients incl. every expert / router tensor: bit for bit; fp32 vector gradients (norm weights, embedding rows) are
                # summed with atomics and differ in the last bit between two runs of the SAME path
                assert torch.equal(got, g) or (g.dtype == torch.float32 and 'experts' not in n and rel_err(got.float(), g.float()) < 1e-5), (mode, n)
        assert float(outs['padded'][0].float().abs().max()) > 0 and len(outs['plain'][2]) >= 25
    finally:
        if own:
            dist.destroy_process_group()


def test_qwen3moe_width_pair_vs_the_reference_trainer():
    """BASELINE configs[4]'s backbone pinned to the reference at FULL WIDTH (round 5).  tests/golden/qwen3moe_width_dpo.npz: the UNMODIFIED text-to-text DPOTrainer
    (trainers/text_to_text/dpo.py:122-203) on oracle.synthetic.qwen3moe_width in the build container -- 2 sparse layers of the Qwen3-30B-A3B geometry with all 128
    experts (top-8 of 128, normalised weights), per-head q / k norms, GQA 32 / 4, the 151936-row head; one left-padded pair -- in fp32 and in the reference's own
    bf16.  The native path: device-side routing + expert-major plan + row-grouped expert GEMMs.  Bounds: tests/width_parity.py."""
    from oracle.synthetic import qwen3moe_width
    from tests.util import load_golden
    from tests.width_parity import width_parity
    z = load_golden('qwen3moe_width_dpo.npz')
    hc, sd, ref_sd, batch, PAD = qwen3moe_width()
    width_parity(z, hc, sd, ref_sd, batch, PAD, 'parity_qwen3moe_width_vs_reference.txt', min_matrices=14)
