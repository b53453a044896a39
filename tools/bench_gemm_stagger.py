"""Diagnostic: does de-synchronising the CUs' epilogues (AA_GEMM_ABLATE) lower the per-tile fixed cost of the persistent gemm4 kernel?
M = 65536 x N = 4096 (16 rounds of 256 tiles), K = 1024 / 4096."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align_anything_amd import ops
dev = torch.device('cuda:0')
def rnd(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
ops.gemm_set_tile(5)
M, N = 65536, 4096
out = {}
for K in (1024, 4096):
    a, b = rnd(M, K), rnd(N, K)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    us = timeit(lambda: ops.gemm(a, b, out=c))
    out[K] = (round(us, 1), round(2.0 * M * N * K / us / 1e6, 1))
print('ablate', os.environ.get('AA_GEMM_ABLATE', '0'), out, flush=True)
