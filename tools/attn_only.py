import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align_anything_amd import ops
dev = torch.device('cuda:0')
N, T, H, hd = 4, 2048, 32, 128
qkv = (torch.randn(N * T, 3 * H * hd, device=dev) * 0.5).to(torch.bfloat16)
q, k, v = qkv[:, :H * hd], qkv[:, H * hd:2 * H * hd], qkv[:, 2 * H * hd:]
for _ in range(3):
    o, lse = ops.attn_fwd(q, k, v, N, T, H, H, hd, True, hd ** -0.5)
do = torch.randn_like(o); dqkv = torch.empty_like(qkv)
for _ in range(2):
    ops.attn_bwd(q, k, v, o, do, lse, dqkv[:, :H * hd], dqkv[:, H * hd:2 * H * hd], dqkv[:, 2 * H * hd:], N, T, H, H, hd, True, hd ** -0.5)
a = (torch.randn(8192, 4096, device=dev) * 0.5).to(torch.bfloat16); b = (torch.randn(4096, 4096, device=dev) * 0.5).to(torch.bfloat16)
for _ in range(3):
    ops.gemm(a, b); ops.gemm(a, b, b_n=True)
torch.cuda.synchronize()
