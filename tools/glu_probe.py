"""Does the time of the fused GLU kernels depend on WHERE their big operands lie relative to each other?  (Same kernel, same data:
1.43 ms in one bench process, 2.29 ms in another -- profiles/r02_*kernel_stats*.csv.)  One arena, gu / dgu (backward) and gu / act
(forward) carved at controlled byte distances; 7B shapes, M = 16384 tokens."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align_anything_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
M, F, K = 16384, 11008, 4096
bf = torch.bfloat16
g = torch.Generator(device='cpu').manual_seed(0)
dy = (torch.randn(M, K, generator=g) * 0.5).to(bf).to(dev)
x = (torch.randn(M, K, generator=g) * 0.5).to(bf).to(dev)
w_down = (torch.randn(K, F, generator=g) * 0.02).to(bf).to(dev)
w_gu = (torch.randn(2 * F, K, generator=g) * 0.02).to(bf).to(dev)
GU_B = M * 2 * F * 2
arena = torch.empty((7 << 30), dtype=torch.uint8, device=dev)
base = (arena.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20) - arena.data_ptr()      # 2 MB-aligned start


def view(off, rows, cols):
    return arena[off:off + rows * cols * 2].view(bf).view(rows, cols)


def timed(fn, reps=8):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return round(best * 1e3, 1)


gu = view(base, M, 2 * F)
gu.copy_((torch.randn(M, 2 * F, generator=g) * 0.5).to(bf))
res = {'bwd': {}, 'fwd': {}}
R2M = (GU_B + (2 << 20) - 1) // (2 << 20) * (2 << 20)
# byte distances dgu - gu: just past the operand, then "round" distances (powers of two and sums of two of them), where a
# channel / set hash that ignores the high address bits would make the two streams of a tile collide
dists = [R2M, 0x40000000, 0x100000000] if os.environ.get('AA_PROBE_SHORT') else \
        [R2M, R2M + 4096, 0x2C000000, 0x30000000, 0x38000000, 0x40000000, 0x40000000 + 4096, 0x50000000, 0x60000000, 0x80000000,
         0x80000000 + (1 << 21), 0xC0000000, 0x100000000, 0x100000000 + 0x2B200000, 0x140000000]
for d in dists:
    off = base + d
    dgu = view(off, M, 2 * F)

    def bwd():
        ops.call('aa_gemm_glu_bwd_bf16', dy.data_ptr(), w_down.data_ptr(), gu.data_ptr(), dgu.data_ptr(), None, M, F, K, dy.stride(0),
                 w_down.stride(0), gu.stride(0), dgu.stride(0), ops.stream())
    res['bwd'][d] = timed(bwd)
    print(f'dgu - gu = {d:#13x}: glu_bwd {res["bwd"][d]:7.1f} us', flush=True)
# does it depend on where dy / w_down lie instead?  (fresh copies at different allocator positions)
for i in range(0 if os.environ.get('AA_PROBE_SHORT') else 4):
    pad = torch.empty(((i + 1) * 97 << 20) + i * 4096, dtype=torch.uint8, device=dev)
    dy2, w2 = dy.clone(), w_down.clone()
    dgu = view(base + R2M, M, 2 * F)
    t = timed(lambda: ops.call('aa_gemm_glu_bwd_bf16', dy2.data_ptr(), w2.data_ptr(), gu.data_ptr(), dgu.data_ptr(), None, M, F, K, dy2.stride(0),
                               w2.stride(0), gu.stride(0), dgu.stride(0), ops.stream()))
    print(f'dy at {dy2.data_ptr():#x} w_down at {w2.data_ptr():#x}: glu_bwd {t:7.1f} us', flush=True)
    del pad
# the allocator's own placement, as the trainer gets it
gu2 = torch.empty((M, 2 * F), dtype=bf, device=dev); gu2.copy_(gu)
dgu2 = torch.empty_like(gu2)
print('torch.empty placement: gu %x dgu %x distance %d' % (gu2.data_ptr(), dgu2.data_ptr(), dgu2.data_ptr() - gu2.data_ptr()),
      'glu_bwd', timed(lambda: ops.call('aa_gemm_glu_bwd_bf16', dy.data_ptr(), w_down.data_ptr(), gu2.data_ptr(), dgu2.data_ptr(), None, M, F, K,
                                        dy.stride(0), w_down.stride(0), gu2.stride(0), dgu2.stride(0), ops.stream())), 'us')
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'glu_probe.json'), 'w'))
