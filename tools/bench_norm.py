"""RMSNorm forward / backward at the shapes of the headline step (rows = 2 B T = 16384 at B = 4 and 4096 at B = 1, h = 4096), HIP-event timed, bytes
counted as the kernels move them: forward reads x, writes y (4 h bytes per row); backward reads x, dy and -- with the residual gradient added, as every
call of the decoder stack does -- dx, writes dx (6 or 8 h bytes per row).  -> gpurun_out/<out>.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from align_anything_amd import ops  # noqa: E402
from tools.bench_kernels import rnd, timeit  # noqa: E402

res = []
for M in (16384, 4096):
    h = 4096
    x, w = rnd(M, h), rnd(h)
    y, rstd = ops.rmsnorm_fwd(x, w, 1e-5)
    ms = timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-5, out=y, rstd=rstd), iters=50)
    res.append(dict(kernel='rmsnorm_fwd', rows=M, h=h, us=ms * 1e3, gbs=2 * M * h * 2 / ms / 1e6))
    dw = torch.zeros(h, device='cuda'); dx = torch.zeros_like(x)
    for add in (False, True):
        ms = timeit(lambda: ops.rmsnorm_bwd(y, x, w, rstd, dw, dx=dx, add_to_dx=add), iters=50)
        res.append(dict(kernel='rmsnorm_bwd' + ('+residual' if add else ''), rows=M, h=h, us=ms * 1e3, gbs=(4 if add else 3) * M * h * 2 / ms / 1e6))
for r in res:
    r['frac_hbm_peak'] = r['gbs'] / 8000.0
    print(r, flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', sys.argv[1] if len(sys.argv) > 1 else 'bench_norm.json'), 'w'), indent=1)
