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
