"""Run ONE GEMM shape a few times with our kernel and with torch.matmul (hipBLASLt) -- meant to be wrapped in rocprofv3
(--kernel-trace for the library's kernel name / duration, --pmc for counters).  args: layout M N K [tile]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align_anything_amd import ops
dev = torch.device('cuda:0')
layout, m, n, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
tile = int(sys.argv[5]) if len(sys.argv) > 5 else -1
a_t, b_n = layout == 'tn', layout in ('nn', 'tn')
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
a = rnd(k, m) if a_t else rnd(m, k)
b = rnd(k, n) if b_n else rnd(n, k)
out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
ops.gemm_set_tile(tile)
for _ in range(4):
    ops.gemm(a, b, out=out, a_t=a_t, b_n=b_n)
torch.cuda.synchronize()
if os.environ.get('AA_PROBE_BLASLT', '1') == '1':
    A = a.t() if a_t else a; B = b if b_n else b.t()
    for _ in range(4):
        torch.matmul(A, B, out=out)
torch.cuda.synchronize()
