"""Same-process A/B of GEMM kernel variants on the 12 hot 7B shapes at M = 16384 tokens (random bf16 operands), with
torch.matmul (hipBLASLt) on the same tensors as a yardstick only.  Variants: env AA_LAB_VARIANTS = comma list of
`name:tile` (tile = aa_gemm_set_tile id; further fields once selected the K-loop schedule variants of rounds 1-2 and the 32x32x16 kernel of tools/lab/gemm5
while it was wired in, commit adb854f).  Writes gpurun_out/gemm_lab.json."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from align_anything_amd import ops
dev = torch.device('cuda:0')
TOK = int(os.environ.get('AA_LAB_TOKENS', 16384))
def rnd(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
def timeit(fn, iters=12, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
variants = []
for v in os.environ.get('AA_LAB_VARIANTS', 'g8:0,g4:5').split(','):
    parts = v.split(':')
    variants.append((parts[0], int(parts[1]), int(parts[2]) if len(parts) > 2 else -1))
shapes = [('qkv', 12288, 4096), ('o', 4096, 4096), ('gate_up', 22016, 4096), ('down', 4096, 11008)]
only = os.environ.get('AA_LAB_LAYOUTS', 'nt,nn,tn').split(',')
res = []
for name, N, K in shapes:
    for layout in only:
        a_t, b_n = layout == 'tn', layout in ('nn', 'tn')
        if layout == 'nn': m, n, k = TOK, K, N
        elif layout == 'tn': m, n, k = N, K, TOK
        else: m, n, k = TOK, N, K
        a = rnd(k, m) if a_t else rnd(m, k)
        b = rnd(k, n) if b_n else rnd(n, k)
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        fl = 2.0 * m * n * k
        row = dict(name=name, layout=layout, m=m, n=n, k=k)
        rows = torch.arange(0, m, 97, device=dev)[:256]
        ref = ((a[:, rows].t() if a_t else a[rows]).float()) @ (b if b_n else b.t()).float()
        for rep in range(2):
            for vn, tile, ilv in variants:
                ops.gemm_set_tile(tile)
                ms = timeit(lambda: ops.gemm(a, b, out=out, a_t=a_t, b_n=b_n))
                row[f'{vn}_tf_{rep}'] = round(fl / ms / 1e9, 1)
                if rep == 0:
                    row[f'{vn}_relerr'] = round((out[rows].float() - ref).abs().max().item() / ref.abs().max().item(), 5)
        ops.gemm_set_tile(-1)
        if os.environ.get('AA_LAB_BLASLT', '1') == '1':
            A = a.t() if a_t else a; B = b if b_n else b.t()
            ms = timeit(lambda: torch.matmul(A, B, out=out))
            row['hipblaslt_tf'] = round(fl / ms / 1e9, 1)
        print(row, flush=True); res.append(row)
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', os.environ.get('AA_LAB_OUT', 'gemm_lab.json')), 'w'), indent=1)
