import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align_anything_amd import ops
dev = torch.device('cuda:0')
a = (torch.randn(8192, 4096, device=dev) * 0.5).to(torch.bfloat16); b = (torch.randn(4096, 4096, device=dev) * 0.5).to(torch.bfloat16)
ops.gemm_set_tile(0)
for mode in (0, 1, 3):
    ops.gemm_set_interleave(mode)
    for _ in range(3):
        ops.gemm(a, b)
torch.cuda.synchronize()
