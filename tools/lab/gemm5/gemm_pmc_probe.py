"""One launch of each big-tile GEMM variant per layout, for a rocprofv3 --pmc pass (LDS bank conflicts / MFMA busy of gemm4 vs gemm5):
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace ... -- python tools/gemm_pmc_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align_anything_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
M, N, K = 8192, 4096, 4096
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
for layout in ('nt', 'nn', 'tn'):
    a_t, b_n = layout == 'tn', layout in ('nn', 'tn')
    a = rnd(K, M) if a_t else rnd(M, K)
    b = rnd(K, N) if b_n else rnd(N, K)
    for m32 in (False, True):
        ops.gemm_set_tile(5)
        ops.gemm_set_mfma32(m32)
        for _ in range(3):
            ops.gemm(a, b, a_t=a_t, b_n=b_n)
        torch.cuda.synchronize()
ops.gemm_set_tile(-1)
ops.gemm_set_mfma32(False)
