"""A/B of the GEMM K-loop schedules (simple vs software-pipelined), interleaved in one process."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align_anything_amd import ops
dev = torch.device('cuda:0')
def rnd(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
res = []
shapes = [('qkv', 8192, 12288, 4096), ('o', 8192, 4096, 4096), ('gate_up', 8192, 22016, 4096), ('down', 8192, 4096, 11008)]

for name, M, N, K in shapes:
    for layout in ('nt', 'nn', 'tn'):
        a_t, b_n = layout == 'tn', layout in ('nn', 'tn')
        # backward shapes: nn: dX[M,K'] = dY[M,N'] W[N',K'] ; tn: dW = dY^T X -> contraction over tokens
        if layout == 'nn': m, n, k = M, K, N
        elif layout == 'tn': m, n, k = N, K, M
        else: m, n, k = M, N, K
        a = rnd(k, m) if a_t else rnd(m, k)
        b = rnd(k, n) if b_n else rnd(n, k)
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        fl = 2.0 * m * n * k
        row = dict(name=name, layout=layout, m=m, n=n, k=k)
        for rep in range(2):
            for ilv in (1, 3):
                ops.gemm_set_tile(0)
                ops.gemm_set_interleave(ilv)
                ms = timeit(lambda: ops.gemm(a, b, out=out, a_t=a_t, b_n=b_n))
                row[f'ilv{ilv}_tf_{rep}'] = round(fl / ms / 1e9, 1)
                if rep == 0 and ilv == 3:
                    ref = (A_ref := (a.t() if a_t else a).float()[:64]) @ (b if b_n else b.t()).float()
                    err = (out[:64].float() - ref).abs().max().item() / ref.abs().max().item()
                    row['ilv3_relerr'] = round(err, 5)
        ops.gemm_set_interleave(-1)
        A = a.t() if a_t else a; B = b if b_n else b.t()
        ms = timeit(lambda: torch.matmul(A, B, out=out))
        row['hipblaslt_tf'] = round(fl / ms / 1e9, 1)
        print(row, flush=True); res.append(row)
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'bench_gemm_ab.json'), 'w'), indent=1)
