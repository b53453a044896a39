"""GPU: flat AdamW + global-norm clip (csrc/optim.hip) vs the oracle restatement of FusedAdam."""
import pytest
import torch

from oracle import rl_math as orl
from tests.gpu_util import assert_close, dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('thin', [False, True])
@pytest.mark.parametrize('gdtype', [torch.bfloat16, torch.float32])
def test_adamw_flat_five_steps(gdtype, thin):
    """thin = the <= 16-VGPR kernel that runs co-resident with the GEMM tiles (hardware sqrt / rcp, same update)."""
    from align_anything_amd import ops
    ops.adamw_set_thin(thin)
    n = 100003
    gen = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=gen)
    master = p0.clone().to(dev()); m = torch.zeros(n, device=dev()); v = torch.zeros(n, device=dev())
    p16 = torch.empty(n, dtype=torch.bfloat16, device=dev())
    q, qm, qv = p0.clone(), torch.zeros(n), torch.zeros(n)
    sumsq = torch.zeros(1, device=dev()); coef = torch.zeros(1, device=dev()); nrm = torch.zeros(1, device=dev())
    for step in range(1, 6):
        g = (torch.randn(n, generator=gen) * 0.01 * step).to(gdtype)
        gd = g.to(dev())
        sumsq.zero_()
        ops.grad_sumsq_(gd, sumsq)
        ops.clip_coef(sumsq, 1.0, coef, nrm)
        ops.adamw_flat_(master, m, v, p16, gd, 1e-3, 0.9, 0.95, 1e-8, 0.05, step, clip=coef)
        c, tot = orl.clip_coef([g], 1.0)
        assert abs(nrm.item() - tot.item()) < 1e-3 * tot.item()
        orl.adamw_step(q, g.float() * c, qm, qv, step, 1e-3, 0.9, 0.95, 1e-8, 0.05)
    torch.cuda.synchronize()
    ops.adamw_set_thin(False)
    assert_close(master.cpu(), q, rtol=1e-5, atol=2e-6, what='master')
    assert torch.equal(p16, master.to(torch.bfloat16)), 'bf16 shadow must be RNE of the fp32 master'


@pytest.mark.parametrize('thin', [False, True])
def test_skip_sentinel_leaves_weights_and_moments_untouched(thin):
    """sumsq == -inf (the all-reduced capacity-overflow flag of the sync-free expert exchange, expert_parallel.py) -> coefficient and
    norm -1 -> the AdamW kernels return without writing: the invalid step never reaches weights or optimizer state."""
    from align_anything_amd import ops
    ops.adamw_set_thin(thin)
    n = 70001
    gen = torch.Generator().manual_seed(1)
    master = torch.randn(n, generator=gen).to(dev()); m = torch.randn(n, generator=gen).to(dev()); v = torch.rand(n, generator=gen).to(dev())
    p16 = master.to(torch.bfloat16)
    keep = [t.clone() for t in (master, m, v, p16)]
    g = torch.randn(n, generator=gen).to(torch.bfloat16).to(dev())
    sumsq = torch.full((1,), float('-inf'), device=dev()); coef = torch.zeros(1, device=dev()); nrm = torch.zeros(1, device=dev())
    ops.clip_coef(sumsq, 1.0, coef, nrm)
    ops.adamw_flat_(master, m, v, p16, g, 1e-3, 0.9, 0.95, 1e-8, 0.05, 3, gscale=0.5, clip=coef)
    torch.cuda.synchronize()
    ops.adamw_set_thin(False)
    assert coef.item() == -1.0 and nrm.item() == -1.0
    for a, b in zip((master, m, v, p16), keep):
        assert torch.equal(a, b)
    # +inf (diverged gradients) is NOT the sentinel: coefficient 0, the update runs as before
    sumsq.fill_(float('inf'))
    ops.clip_coef(sumsq, 1.0, coef, nrm)
    assert coef.item() == 0.0 and nrm.item() == float('inf')


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('world', [2, 8])
def test_chunk_sum_is_the_rank_ordered_fp32_sum_rounded_once(dtype, world):
    """aa_chunk_sum (reduce step of the direct gradient exchange, engine.GradReducer): out = sum_r in[r] accumulated in fp32 in rank order and rounded
    once to the gradient dtype -- bit-exact against the same sum in torch."""
    from align_anything_amd import ops
    c = 8 * 12345
    g = torch.Generator().manual_seed(3)
    recv = (torch.randn(world * c, generator=g) * 1e-3).to(dtype).to(dev())
    out = torch.empty(c, dtype=dtype, device=dev())
    ops.chunk_sum(recv, out, world)
    acc = torch.zeros(c, dtype=torch.float32, device=dev())
    for r in range(world):
        acc = acc + recv[r * c:(r + 1) * c].float()
    torch.cuda.synchronize()
    assert torch.equal(out, acc.to(dtype))
