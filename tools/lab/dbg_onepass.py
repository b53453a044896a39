import sys, torch
sys.path.insert(0, '/root/repo')
from align_anything_amd import ops
dev = torch.device('cuda:0')
def run(N, T, H, Hkv, causal=True):
    hd = 128
    g = torch.Generator().manual_seed(0)
    W = (H + 2 * Hkv) * hd
    qkv = (torch.randn(N * T, W, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:(H + Hkv) * hd], qkv[:, (H + Hkv) * hd:]
    do = (torch.randn(N * T, H * hd, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    sc = hd ** -0.5
    o, lse = ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, causal, sc)
    outs = {}
    for name, flag in (('two', False), ('one', True)):
        ops.ATTN_ONEPASS = flag
        d = torch.full_like(qkv, float('nan'))
        dq, dk, dv = d[:, :H * hd], d[:, H * hd:(H + Hkv) * hd], d[:, (H + Hkv) * hd:]
        ops.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, N, T, H, Hkv, hd, causal, sc)
        torch.cuda.synchronize()
        outs[name] = d.float()
    a, b = outs['one'], outs['two']
    print(f'--- N{N} T{T} H{H}/{Hkv} causal {causal}: nan in one: {int(torch.isnan(a).sum())}')
    for nm, lo, hi in (('dq', 0, H * hd), ('dk', H * hd, (H + Hkv) * hd), ('dv', (H + Hkv) * hd, W)):
        x, y = a[:, lo:hi], b[:, lo:hi]
        print(nm, 'one: nonzero frac', float((x != 0).float().mean()), 'absmax', float(x.abs().max()), '| two absmax', float(y.abs().max()), '| max diff', float((x - y).abs().max()))
        # per 64-row block of the first sequence: relative error
        blocks = []
        for r0 in range(0, min(T, 512), 64):
            xx, yy = x[r0:r0 + 64], y[r0:r0 + 64]
            blocks.append(round(float((xx - yy).norm() / (yy.norm() + 1e-9)), 3))
        print('   rel err per 64-row block (seq 0):', blocks)
        # per 32-column block of head 0
        cols = [round(float((x[:T, c0:c0 + 32] - y[:T, c0:c0 + 32]).norm() / (y[:T, c0:c0 + 32].norm() + 1e-9)), 3) for c0 in range(0, 128, 32)]
        print('   rel err per 32-col block (head 0):', cols)
run(1, 128, 1, 1)
run(1, 256, 1, 1)
run(1, 256, 1, 1, causal=False)
run(2, 512, 4, 2)
