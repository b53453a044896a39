"""Fixed per-tile cost vs per-K-tile cost of a GEMM kernel: time(K) at M=16384, N=4096 (1024 tiles = 4 rounds of 256) for
K = 64 ... 8192; a least-squares line time = a + b * (K / 64) gives a / 4 = fixed cost of one tile round (launch + prologue +
epilogue) and b / 4 = cost of one K-tile iteration.  Variants as in bench_gemm_lab.py."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align_anything_amd import ops
dev = torch.device('cuda:0')
def rnd(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us
variants = []
for v in os.environ.get('AA_LAB_VARIANTS', 'base:0,g4:5').split(','):
    parts = v.split(':'); variants.append((parts[0], int(parts[1]), int(parts[2]) if len(parts) > 2 else -1))
M, N = 16384, 4096
out = {}
for vn, tile, ilv in variants:
    rows = []
    for K in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
        a, b = rnd(M, K), rnd(N, K)
        c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm_set_tile(tile); ops.gemm_set_interleave(ilv)
        us = timeit(lambda: ops.gemm(a, b, out=c))
        rows.append((K, round(us, 2)))
    ops.gemm_set_tile(-1); ops.gemm_set_interleave(-1)
    xs = [k / 64 for k, _ in rows[3:]]; ys = [u for _, u in rows[3:]]
    n = len(xs); mx, my = sum(xs) / n, sum(ys) / n
    b_ = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs); a_ = my - b_ * mx
    out[vn] = {'us_by_K': rows, 'fixed_us_per_round': round(a_ / 4, 2), 'us_per_ktile_per_round': round(b_ / 4, 3),
               'ideal_us_per_ktile_at_2.4GHz': round(128 * 16 / 2400, 3)}
    print(vn, out[vn], flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'gemm_ksweep.json'), 'w'), indent=1)
