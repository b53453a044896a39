"""AdamW on a flat 1.5 G-element group (bf16 gradients, fp32 master / moments, bf16 shadow: 28 B per element = 42 GB per launch), HIP-event timed.
    python tools/bench_adam.py [out.json]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from align_anything_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

dev = torch.device('cuda:0')
n = 1536 * 1024 * 1024
master = torch.zeros(n, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
p16 = torch.zeros(n, dtype=torch.bfloat16, device=dev); g = torch.full((n,), 1e-3, dtype=torch.bfloat16, device=dev)
coef = torch.ones(1, device=dev)
ms = timeit(lambda: ops.adamw_flat_(master, m, v, p16, g, 1e-6, 0.9, 0.95, 1e-8, 0.0, 1, 1.0, coef), iters=8, warm=2)
r = {'kernel': 'adamw_flat', 'n': n, 'ms': ms, 'gbs': 28.0 * n / ms / 1e6, 'frac_hbm_peak': 28.0 * n / ms / 1e6 / 8000.0, 'ms_at_6.76e9_params': ms * 6.76e9 / n}
print(json.dumps(r), flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(r, open(os.path.join(ROOT, 'gpurun_out', sys.argv[1] if len(sys.argv) > 1 else 'bench_adam.json'), 'w'))
