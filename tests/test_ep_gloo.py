"""CPU, world_size 2 and 4, gloo: the token exchange of the expert-parallel MoE block (align_anything_amd/expert_parallel.py) -- counts,
dispatch to the ranks that own the experts, local expert ids of the arriving rows, and the mirrored return path.  The device
kernels are replaced by their integer definition (a stable sort by expert = `aa_moe_plan` with align 1) so the bookkeeping of
the exchange itself is pinned without a GPU; tests/test_ep_gpu.py runs the real block."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from align_anything_amd.expert_parallel import ExpertParallel
    ep = ExpertParallel(dist.new_group())
    ok = ep.size == world and ep.rank == rank and ep.host_staged
    E, k, h = 8, 2, 6
    e0, El = ep.local_experts(E)
    ok = ok and (e0, El) == (rank * (E // world), E // world)
    for trial, M in enumerate((13, 1, 40)):
        g = torch.Generator().manual_seed(100 * trial + rank)
        idx = torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(M)])        # [M, k] distinct experts per token
        if trial == 1 and rank == 0:
            idx = torch.tensor([[E - 3, E - 2]])                                           # rank 0 sends nothing to itself
        flat = idx.reshape(-1)
        order = torch.argsort(flat, stable=True)                                           # dense expert-major send order
        tok, choice = order // k, order % k
        xs = torch.stack([torch.full_like(tok, rank), tok, choice, flat[order], torch.zeros_like(tok), torch.zeros_like(tok)], 1).float()
        counts = torch.bincount(flat, minlength=E).to(torch.int32)
        send, recv, recv_counts = ep.exchange_counts(counts)
        ok = ok and send == [int(counts[r * El:(r + 1) * El].sum()) for r in range(world)] and sum(send) == M * k
        xr = ep.exchange_rows(xs, send, recv)
        ids = ExpertParallel.local_expert_ids(recv_counts, 'cpu').view(-1).long()
        ok = ok and xr.shape[0] == sum(recv) == ids.numel()
        ok = ok and bool((xr[:, 3].long() == e0 + ids).all())                              # every row reached the owner of its expert
        src_rank = torch.repeat_interleave(torch.arange(world), torch.tensor(recv))
        ok = ok and bool((xr[:, 0].long() == src_rank).all())                              # source-rank major
        # within one (source, expert) block the sender's token order is kept (stable)
        key = xr[:, 0] * 1e6 + xr[:, 3] * 1e3 + xr[:, 1]
        ok = ok and bool((key[1:] >= key[:-1]).all())
        ys = ep.exchange_rows(xr * 2.0, recv, send)                                        # "expert output" travels back
        ok = ok and torch.equal(ys, xs * 2.0)
        # bf16 rows travel as raw words
        xb = xs.to(torch.bfloat16)
        ok = ok and torch.equal(ep.exchange_rows(ep.exchange_rows(xb, send, recv), recv, send), xb)
    full = ep.all_gather_rows(torch.full((El, 3), float(rank)).to(torch.bfloat16))
    ok = ok and full.shape == (E, 3) and full[:, 0].tolist() == [float(r) for r in range(world) for _ in range(El)]
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


# ---- the integer definitions of the device kernels the MoE block runs around the exchange (csrc/moe.hip)
def _plan(idx, E):
    """aa_moe_plan, align 1: counts [E], src [pairs] (token of every dense row), pos [rows, k] (dense row of every pair; -1 = no expert)."""
    rows, k = idx.shape
    flat = idx.reshape(-1).long()
    valid = (flat >= 0) & (flat < E)
    order = torch.argsort(torch.where(valid, flat, E), stable=True)[:int(valid.sum())]
    pos = torch.full((rows * k,), -1, dtype=torch.int32)
    pos[order] = torch.arange(order.numel(), dtype=torch.int32)
    src = torch.full((rows * k,), -1, dtype=torch.int32)
    src[:order.numel()] = (order // k).to(torch.int32)
    return {'counts': torch.bincount(flat[valid], minlength=E).to(torch.int32), 'src': src, 'pos': pos.view(rows, k)}


def _gather(x, src):
    out = x[src.long().clamp(min=0)].clone()
    out[src < 0] = 0
    return out


def _combine(y, pos, rows):
    out = torch.zeros((rows, y.shape[1]), dtype=y.dtype)
    for j in range(pos.shape[1]):
        p = pos[:, j].long()
        out += torch.where((p >= 0)[:, None], y[p.clamp(min=0)], torch.zeros_like(out))
    return out


def _padded_worker(rank, world, port, q):
    """The sync-free capacity-padded exchange == the exact one: same rows in the same order in front of every local expert, same combined
    outputs on the way back; a block that does not fit raises at the poll and loses only the rows beyond the capacity."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from align_anything_amd.expert_parallel import ExpertParallel
    exact = ExpertParallel(dist.new_group())
    # world 8 = one expert per rank: with k = 2 distinct experts per token no expert can get more than half of all pairs, so factor 4 (blocks of
    # pairs / 2 rows) can never overflow there; at factor 2 a random router over 8 single-expert ranks does overflow small batches -- and the poll
    # reports it, which is the behaviour the last part of this worker pins
    factor = 2.0 if world <= 4 else 4.0
    ep = ExpertParallel(dist.new_group(), capacity_factor=factor, dense_below=0)
    ok = ep.padded and not exact.padded and ep.capacity(80) == -(-int(factor * 80) // world) and ExpertParallel(exact.group, 2.0).capacity(80) == 80
    E, k, h = 8, 2, 6
    e0, El = ep.local_experts(E)
    # the last two trials give every rank a DIFFERENT number of rows (data-parallel ranks pad to their own batch maximum, rollouts have their own
    # prompt lengths): the block size must come from the agreed maximum (local M made the equal-split all-to-all abort / corrupt)
    for trial, M in enumerate((13, 1, 40, 64, 9 + 11 * rank, 64 * (1 + (world - 1 - rank) % 3))):
        g = torch.Generator().manual_seed(100 * trial + rank)
        idx = torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(M)]).to(torch.int32)
        x = torch.randn(M, h, generator=g)
        lay = _plan(idx, E)
        # exact exchange (the path pinned by test_expert_parallel_exchange)
        send, recv, recv_counts = exact.exchange_counts(lay['counts'])
        xr = exact.exchange_rows(_gather(x, lay['src']), send, recv)
        lp = _plan(ExpertParallel.local_expert_ids(recv_counts, 'cpu'), El)
        xp = _gather(xr, lp['src'])                                               # rows in front of the local experts, expert-major
        fn = lambda t, e: t * (1.0 + e.float()[:, None])                          # "expert e" = scale by 1 + e
        le = torch.repeat_interleave(torch.arange(El), lp['counts'].long())
        ys = exact.exchange_rows(_combine(fn(xp, le), lp['pos'], xr.shape[0]), recv, send)
        want = _combine(ys, lay['pos'], M)
        # padded exchange
        shared = ep.shared_pairs(M * k)
        mx = torch.tensor([M * k]); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        ok = ok and shared == int(mx)
        with ep.pass_scope(M * k):                                                # inside a scope the agreement is made once ...
            ok = ok and ep._scope == (M * k, shared) and ep.shared_pairs(M * k) == shared
        ok = ok and ep._scope is None
        cap = ep.capacity(shared)
        send_src, pos_p = ep.padded_send_layout(lay['counts'], lay['src'], lay['pos'], idx, cap)
        ok = ok and send_src.numel() == world * cap and int((send_src >= 0).sum()) == M * k and bool((pos_p >= 0).all())
        xr2 = ep.exchange_fixed(_gather(x, send_src))
        rc = ep.exchange_counts_device(lay['counts'])
        ok = ok and torch.equal(rc.long(), recv_counts)
        ids = ep.padded_recv_ids(rc, cap)
        ok = ok and ids.shape == (world * cap, 1) and int((ids >= 0).sum()) == sum(recv)
        lp2 = _plan(ids, El)
        ok = ok and torch.equal(lp2['counts'], lp['counts'])
        xp2 = _gather(xr2, lp2['src'])[:xp.shape[0]]
        ok = ok and torch.equal(xp2, xp)                                          # same rows, same order, in front of every local expert
        ys2 = ep.exchange_fixed(_combine(fn(_gather(xr2, lp2['src']), torch.cat([le, torch.zeros(world * cap - le.numel(), dtype=le.dtype)])),
                                         lp2['pos'], world * cap))
        ok = ok and torch.equal(_combine(ys2, pos_p, M), want)
        ep.poll_overflow(block=True)                                              # nothing overflowed: no raise
    # overflow: every rank routes everything to rank 0's experts; factor 1 -> capacity = pairs / world < pairs
    tight = ExpertParallel(ep.group, capacity_factor=1.0, dense_below=0)
    M = 16
    idx = torch.stack([torch.arange(M) % El, (torch.arange(M) + 1) % El], 1).to(torch.int32) if El > 1 else torch.zeros((M, k), dtype=torch.int32)
    lay = _plan(idx, E)
    cap = tight.capacity(M * k)
    send_src, pos_p = tight.padded_send_layout(lay['counts'], lay['src'], lay['pos'], idx, cap)
    ok = ok and cap == M * k // world and int((pos_p >= 0).sum()) == cap and int((send_src >= 0).sum()) == cap
    raised = False
    try:
        tight.poll_overflow(block=True)
    except RuntimeError as e:
        raised = 'capacity' in str(e)
    ok = ok and raised
    tight.poll_overflow(block=True)                                                # the flag was consumed
    # with an engine the flag is SHARED: only the last rank overflows, the -inf sentinel rides the squared-norm all-reduce (engine.step) and every
    # rank raises in the same step -- nobody is left in a collective
    if rank == world - 1:
        tight.padded_send_layout(lay['counts'], lay['src'], lay['pos'], idx, cap)
    sumsq = torch.full((1,), 3.0 + rank)
    sumsq += tight.overflow_sentinel('cpu')
    ok = ok and tight._overflow is None and (bool(torch.isinf(sumsq)) == (rank == world - 1))
    dist.all_reduce(sumsq)
    tight.watch_shared(sumsq)
    raised = False
    try:
        tight.poll_overflow(block=True)
    except RuntimeError as e:
        raised = 'skipped' in str(e)
    ok = ok and raised and float(sumsq) == float('-inf')
    sumsq = torch.full((1,), 3.0) + tight.overflow_sentinel('cpu')                 # a clean step: finite norm, nothing pending, no raise
    dist.all_reduce(sumsq)
    tight.watch_shared(sumsq)
    tight.poll_overflow(block=True)
    ok = ok and bool(torch.isfinite(sumsq))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def _engine_step_worker(rank, world, port, q, first_overflows=False):
    """NativeEngine.step() itself, with one rank overflowing.  step() polls BEFORE it folds the local flag into the squared-norm
    all-reduce; the poll must not eat that flag.  The three optimizer kernels are replaced by their contract on CPU tensors (csrc/optim.hip:
    sumsq == -inf -> coefficient -1 -> AdamW returns without touching anything); everything else -- the poll, the sentinel, the all-reduce, the
    shared watch -- is the engine's own code.  Every rank must skip the update and every rank must raise at its next poll."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from types import SimpleNamespace
    from align_anything_amd import engine as eng, ops
    from align_anything_amd.expert_parallel import ExpertParallel
    from align_anything_amd.params import ParamStore
    touched = []

    def grad_sumsq_(g, out, scale=1.0, ws=None):
        out += (g.float() * scale).pow(2).sum()

    def clip_coef(sumsq, max_norm, coef, norm=None):
        if float(sumsq) == float('-inf'):
            coef.fill_(-1.0)
            norm.fill_(-1.0)
        else:
            norm.copy_(sumsq.sqrt())
            coef.fill_(min(1.0, max_norm / (float(norm) + 1e-6)) if max_norm else 1.0)

    def adamw_flat_(master, m, v, p16, g, lr, b1, b2, eps, wd, step, gscale=1.0, clip=None):
        if gscale * float(clip) < 0:
            return
        touched.append(step)
        master -= lr * g.float() * gscale * float(clip)

    ops.grad_sumsq_, ops.clip_coef, ops.adamw_flat_, ops.adamw_set_thin = grad_sumsq_, clip_coef, adamw_flat_, lambda on: None
    st = ParamStore('cpu', torch.float32)
    st.add('experts.w', (8 // world, 3), trainable=True, shard=(rank * (8 // world), 8))
    st.add('dense.w', (3, 4), trainable=True)
    st.allocate()
    ep = ExpertParallel(dist.new_group(), capacity_factor=1.0, dense_below=0)
    module = SimpleNamespace(store=st, ep=ep, device=torch.device('cpu'), init_training=st.init_training)
    e = eng.NativeEngine(module, lr=0.1, lr_scheduler_type='constant', weight_decay=0.0)
    ok = e.world == world and ep.padded

    def one_step(overflow_here):
        for g in st.gflat.values():
            g.fill_(0.5)
        if overflow_here:             # what padded_send_layout does when a block does not fit
            ep._overflow = torch.tensor(True)
        e.micro_steps += 1
        e.step()

    if first_overflows:
        # the engine's VERY FIRST step overflows on one rank, after a stand-alone poll (what a PPO / GRPO rollout before the first update does).
        # The engine owns the flag since its construction, so neither that poll nor step()'s leading one may eat it.
        ok = ok and ep._engine
        before = {g: t.clone() for g, t in st.master.items()}
        if rank == world - 1:
            ep._overflow = torch.tensor(True)
        ep.poll_overflow(block=True)                                   # a rollout's poll: reads only the (empty) shared flag, raises nothing, consumes nothing
        ok = ok and ((ep._overflow is not None) == (rank == world - 1))
        one_step(False)                                                # the flag set above is still there for the sentinel
        ok = ok and float(e._coef) == -1.0 and not touched and all(torch.equal(before[g], st.master[g]) for g in before)
        raised = False
        try:
            e.grad_norm()
        except RuntimeError as err:
            raised = 'skipped' in str(err)
        ok = ok and raised
    before = {g: t.clone() for g, t in st.master.items()}
    one_step(False)                                                    # a clean step first: the poll inside the NEXT step has a landed, clear flag to consume
    ok = ok and touched and float(e._coef) > 0 and all(not torch.equal(before[g], st.master[g]) for g in before)
    before = {g: t.clone() for g, t in st.master.items()}
    n = len(touched)
    one_step(rank == world - 1)                                        # only the last rank overflows
    ok = ok and float(e._coef) == -1.0 and float(e._gnorm) == -1.0 and len(touched) == n           # skipped on EVERY rank
    ok = ok and all(torch.equal(before[g], st.master[g]) for g in before)
    raised = False
    try:
        e.grad_norm()
    except RuntimeError as err:
        raised = 'skipped' in str(err)
    ok = ok and raised                                                 # ... and every rank raises, in the same step
    one_step(False)                                                    # the flags were consumed: training can go on (a caller that catches the error)
    ok = ok and float(e._coef) > 0 and len(touched) > n
    e.grad_norm()
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,first_overflows', [(2, False), (4, False), (2, True)])
def test_engine_step_shares_the_overflow_flag(world, first_overflows):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000 + world + 10 * first_overflows
    procs = [ctx.Process(target=_engine_step_worker, args=(r, world, port, q, first_overflows)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, True) for r in range(world)]


@pytest.mark.parametrize('world', [2, 4, 8])
def test_capacity_padded_exchange_equals_exact_exchange(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_padded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, True) for r in range(world)]


@pytest.mark.parametrize('world', [2, 4])
def test_expert_parallel_exchange(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, True) for r in range(world)]


def test_sharded_blocks_load_their_rows_of_a_full_checkpoint():
    from align_anything_amd.params import ParamStore
    st = ParamStore('cpu', torch.float32)
    st.add('experts.w', (2, 3, 4), trainable=True, shard=(2, 8))
    st.add('dense.w', (3, 4), trainable=True)
    st.allocate()
    assert st.specs['experts.w']['group'] == 'exp' and st.specs['dense.w']['group'] == 'mat'
    full = torch.arange(8 * 3 * 4, dtype=torch.float32).view(8, 3, 4)
    st.load_state_dict({'experts.w': full, 'dense.w': torch.ones(3, 4)})
    assert torch.equal(st.p['experts.w'], full[2:4])
    st.load_state_dict({'experts.w': full[2:4] + 1, 'dense.w': torch.ones(3, 4)})          # an already sharded tensor loads as is
    assert torch.equal(st.p['experts.w'], full[2:4] + 1)
    st.init_training()
    assert st.trainable_groups() == ['mat', 'exp'] and st.gflat['exp'].numel() >= 24
