"""The CLIP-L/14-336 tower's GEMM shapes at the DPO step's batch (4 images: M = 4 x 577 = 2308 rows), per tile configuration of csrc/gemm.hip
(aa_gemm_set_tile: -1 heuristic, 0 = 256x256, 1 = 128x128, 2 = 256x128, 3 = 128x256) -- which tile each shape should take.
    python tools/bench_clip_gemms.py  -> one line per (shape, tile)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align_anything_amd import ops  # noqa: E402
from align_anything_amd.lib import LIB  # noqa: E402

dev = torch.device('cuda:0')
LIB.load()
M = int(os.environ.get('CLIP_M', 2308))
shapes = {'patch_embed (K 640)': (1024, 640, False, 0, False), 'qkv': (3072, 1024, True, 0, False), 'out_proj + residual': (1024, 1024, True, 0, True),
          'fc1 + quick_gelu': (4096, 1024, True, ops.ACT_QUICK_GELU, False), 'fc2 + residual': (1024, 4096, True, 0, True)}
res = []
for name, (N, K, bias, act, resid) in shapes.items():
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).to(dev)
    b = torch.zeros(N, dtype=torch.bfloat16, device=dev) if bias else None
    r = (torch.randn(M, N, generator=g) * 0.5).to(torch.bfloat16).to(dev) if resid else None
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    row = {'shape': name, 'M': M, 'N': N, 'K': K}
    for tile in (-1, 0, 1, 2, 3):
        LIB.call('aa_gemm_set_tile', tile)
        try:
            for _ in range(3):
                ops.gemm(x, w, out=out, bias=b, residual=r, act=act)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm(x, w, out=out, bias=b, residual=r, act=act)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            row[f'tile{tile}_us'] = round(us, 1)
            row[f'tile{tile}_tflops'] = round(2.0 * M * N * K / us / 1e6, 1)
        except Exception as ex:
            row[f'tile{tile}_us'] = repr(ex)[:80]
    LIB.call('aa_gemm_set_tile', -1)
    print(json.dumps(row), flush=True)
    res.append(row)
os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out'), exist_ok=True)
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'r04_clip_gemms.json'), 'w'), indent=1)
