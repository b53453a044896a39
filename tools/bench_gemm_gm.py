"""A/B of the tile-group height GM of the L2-aware tile order (interleaved in one process), forward NT + backward shapes."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align_anything_amd import ops
from align_anything_amd.lib import call
dev = torch.device('cuda:0')
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
def timeit(fn, iters=15, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
M = 16384
for name, N, K in (('qkv', 12288, 4096), ('o', 4096, 4096), ('gate_up', 22016, 4096), ('down', 4096, 11008)):
    for layout in ('nt', 'nn', 'tn'):
        a_t, b_n = layout == 'tn', layout in ('nn', 'tn')
        if layout == 'nn': m, n, k = M, K, N
        elif layout == 'tn': m, n, k = N, K, M
        else: m, n, k = M, N, K
        a = rnd(k, m) if a_t else rnd(m, k); b = rnd(k, n) if b_n else rnd(n, k)
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        row = {'name': name, 'layout': layout}
        for rep in range(2):
            for gm in (3, 4, 8, 256 + 2, 256 + 4, 256 + 8):
                call('aa_gemm_set_group', gm)
                ms = timeit(lambda: ops.gemm(a, b, out=out, a_t=a_t, b_n=b_n))
                row[f'gm{gm}_{rep}'] = round(2.0 * m * n * k / ms / 1e9)
        call('aa_gemm_set_group', 0)
        print(row, flush=True)
