"""CPU, world_size 2, gloo: the data-parallel exchange of the native engine (GradReducer buckets + the
fused metric all-reduce) reproduces averaged gradients / metrics."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from align_anything_amd.engine import GradReducer
    from align_anything_amd.trainers.common import get_all_reduce_max, get_all_reduce_mean
    torch.manual_seed(rank)
    flat = torch.randn(1000)
    local = flat.clone()
    red = GradReducer()
    assert red.world == world
    # three "layer" buckets in reverse order + the remainder, as NativeEngine.backward issues them
    for lo, hi in ((700, 1000), (300, 700), (0, 300)):
        red.reduce_async(flat[lo:hi])
    red.wait()
    gathered = [torch.zeros(1000) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered)
    ok = torch.allclose(flat, want, atol=1e-6)
    m = get_all_reduce_mean(torch.tensor([float(rank), 2.0 * rank]))
    mx = get_all_reduce_max(torch.tensor([float(rank)]))
    ok = ok and m.tolist() == [0.5, 1.0] and mx.item() == 1.0
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_bucketed_grad_allreduce_and_metric_reduce_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def _worker_bf16(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from align_anything_amd.engine import GradReducer
    n = 1 << 20
    # realistic gradient statistics: a shared signal whose magnitudes spread over two decades around 1e-4 (what the flat bf16 'mat'
    # gradient of a DPO step looks like) + per-rank data noise of the same order -- every rank's slice is a bf16 tensor, as in the engine
    gs = torch.Generator().manual_seed(1234)
    signal = torch.randn(n, generator=gs) * torch.pow(10.0, -4.0 + torch.randn(n, generator=gs).clamp(-2, 2) * 0.5)
    gr = torch.Generator().manual_seed(99 + rank)
    mine = (signal * (1.0 + 0.5 * torch.randn(n, generator=gr)) + 0.7 * signal.abs() * torch.randn(n, generator=gr)).to(torch.bfloat16)
    flat = mine.clone()
    red = GradReducer()
    for lo, hi in ((700000, n), (300000, 700000), (0, 300000)):          # per-layer buckets, last layer first (NativeEngine.backward)
        red.reduce_async(flat[lo:hi])
    red.wait()
    parts = [torch.zeros(n, dtype=torch.bfloat16) for _ in range(world)]
    dist.all_gather(parts, mine)
    exact = sum(p.double() for p in parts)                               # the fp32/fp64 sum of the SAME bf16 inputs (DeepSpeed's fp32 option)
    got = flat.double()
    ulp = 2.0 ** -8
    # rounding happens on the partial sums of the ring, whose size is bounded by sum_r |g_r| (an element whose ranks cancel keeps the
    # error of its large partial sums): max error in ulps of that bound; rms in ulps of the result where it did not cancel
    mag = sum(p.double().abs() for p in parts)
    err_max = ((got - exact).abs() / (mag * ulp).clamp_min(1e-300)).max()
    solid = exact.abs() >= 0.5 * mag
    err = ((got - exact).abs() / (exact.abs() * ulp).clamp_min(1e-300))[solid]
    norm_rel = abs(float(got.norm() / exact.norm()) - 1.0)
    # what reaches the weights: Adam's first update is lr * g / (|g| + eps) = lr * sign(g) -> count the sign flips; and the clipped direction
    flips = float(((got * exact) < 0).double().mean())
    cos = float((got * exact).sum() / (got.norm() * exact.norm()))
    q.put((rank, float(err_max), float(err.pow(2).mean().sqrt()), norm_rel, flips, cos))
    dist.destroy_process_group()


def test_bf16_bucket_allreduce_world8_is_within_bf16_ulps_of_the_fp32_sum():
    """SURVEY section 8(a') 'parity unpinned': DeepSpeed's gradient-communication dtype is a config option the reference's yaml leaves at
    the default; the native engine all-reduces the bf16 'mat' gradient buckets as bf16 SUMs (1/world folded into the optimizer).  At the
    8 ranks of BASELINE's node, with gradient-like magnitudes, the bf16 sum stays within a few bf16 ulps of the fp32 sum of the same
    per-rank tensors: rms error <= 1.5 ulp of the result (elements that do not cancel across ranks), max <= 4 ulp of sum_r |g_r| (the size of the partial sums), global norm (the clip coefficient) to 1e-3, < 0.5 % sign flips (all of them on
    elements that cancel to ~0 across ranks), direction cosine > 0.99999.  All ranks hold the SAME reduced bits (replica consistency)."""
    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_bf16, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    assert [r[0] for r in res] == list(range(world))
    assert len({r[1:] for r in res}) == 1, f'ranks disagree on the reduced gradient: {res}'
    _, mx, rms, norm_rel, flips, cos = res[0]
    print(f'bf16 all-reduce, world 8: max {mx:.2f} ulp, rms {rms:.2f} ulp, |norm ratio - 1| {norm_rel:.2e}, sign flips {100 * flips:.3f} %, cos {cos:.7f}')
    assert mx <= 4.0 and rms <= 1.5 and norm_rel < 1e-3 and flips < 5e-3 and cos > 0.99999, res[0]


def _worker_direct(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from align_anything_amd.engine import GradReducer
    ok = True
    for dtype in (torch.float32, torch.bfloat16):
        n = 64 * world * 37
        g = torch.Generator().manual_seed(7 + rank)
        mine = (torch.randn(n, generator=g) * 1e-3).to(dtype)
        parts = [torch.zeros(n, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(parts, mine.float())
        exact = sum(p.double() for p in parts)
        mag = sum(p.double().abs() for p in parts)                                  # elements whose ranks cancel keep the rounding of their partial sums
        red = GradReducer(mode='direct')
        flat = mine.clone()
        for lo, hi in ((64 * world * 30, n), (64 * world * 11, 64 * world * 30), (0, 64 * world * 11)):      # buckets in the engine's order
            red.reduce_async(flat[lo:hi])
        red.wait()
        # one rounding of the fp32 sum (rank order): exact in fp32 up to summation order, half an ulp in bf16
        tol = 2.0 ** -22 * world if dtype == torch.float32 else 2.0 ** -7.5
        ok = ok and bool(((flat.double() - exact).abs() <= tol * mag + 1e-12).all())
        allf = [torch.zeros(n, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(allf, flat.float())
        ok = ok and all(torch.equal(a, allf[0]) for a in allf)                     # replicas hold identical bits
        ragged = mine[:100].clone()                                                 # a slice the chunking does not fit falls back to all_reduce
        red.reduce_async(ragged); red.wait()
        ok = ok and bool(((ragged.double() - exact[:100]).abs() <= 2.0 ** -6 * mag[:100] + 1e-9).all())
    auto = GradReducer(mode='auto')
    buf = torch.full((64 * world * 4,), float(rank + 1))
    auto.reduce_async(buf); auto.wait()
    rep = auto.autotune_report
    ok = ok and auto.mode in ('ring', 'direct') and rep['chosen'] == auto.mode and rep['forms_agree'] and bool((buf == world * (world + 1) / 2).all())
    modes = [None] * world
    dist.all_gather_object(modes, auto.mode)
    ok = ok and len(set(modes)) == 1                                               # every rank took the same decision
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_direct_gradient_exchange_equals_the_sum_and_keeps_replicas_identical(world):
    """engine.GradReducer mode 'direct' (all-to-all + rank-ordered fp32 sum + all-gather over the point-to-point links) and the
    in-run choice between it and RCCL's all-reduce ('auto'): every bucket equals the exact sum to one rounding, all ranks hold the same bits, a
    ragged slice falls back to all_reduce, and all ranks choose the same form."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33100 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_worker_direct, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, True) for r in range(world)]


def _autotune_one_rank_fails_worker(rank, world, port, q):
    """a form of the gradient exchange that fails on ONE rank only (its workspace allocation).  Before the fix that rank skipped the form
    while its peers entered its collectives -- different collective sequences, a hang.  Now the allocation runs first and the ranks agree on it (MIN) before any
    of them enters the form: everybody drops it, nobody is left inside a collective, and the bucket is still reduced correctly."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))     # a desynchronised sequence would time out here, not hang the suite
    from align_anything_amd.engine import GradReducer
    red = GradReducer(mode='auto')
    real, real_ws = red._direct, red._workspace
    calls = {'n': 0}

    def counted(flat):
        calls['n'] += 1
        return real(flat)

    def flaky_ws(dtype, device, n):           # what CAN fail on one rank alone: the form's workspace allocation (out of memory)
        if rank == world - 1:
            raise RuntimeError('out of memory (simulated) while allocating the direct form\'s workspace')
        return real_ws(dtype, device, n)

    red._direct, red._workspace = counted, flaky_ws
    torch.manual_seed(rank)
    flat = torch.randn(4096)
    local = flat.clone()
    red.reduce_async(flat)            # first bucket: autotune (collective), then the exchange in the agreed form
    red.wait()
    gathered = [torch.zeros(4096) for _ in range(world)]
    dist.all_gather(gathered, local)
    ok = red.mode == 'ring' and torch.allclose(flat, sum(gathered), atol=1e-6)
    rep = red.autotune_report or {}
    ok = ok and rep.get('chosen') == 'ring' and (('error_direct' in rep) == (rank == world - 1))
    # nobody entered the direct form's collectives: the allocation's failure was agreed on first
    ok = ok and calls['n'] == 0
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_autotune_survives_a_form_that_fails_on_one_rank(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 37500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_autotune_one_rank_fails_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, True) for r in range(world)]
