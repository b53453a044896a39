"""In-kernel clocks of gemm4 (timing build: AA_HIPCC_EXTRA=-DAA_G4_TIMING).  Wave 0 of every workgroup stamps s_memtime (shader clock) and
s_memrealtime (100 MHz) at kernel entry, K-loop entry, K-loop exit and after the epilogue's stores; this prints, per shape, the shader
clock the chip actually ran at, cycles per K-tile (2048 = the wave's 128 MFMAs back to back), prologue and epilogue cycles."""
import ctypes, json, sys
import numpy as np
import torch
from align_anything_amd import ops
from align_anything_amd.lib import LIB

dll = LIB.load()
dump = dll.aa_gemm4_timing_dump
dump.restype = ctypes.c_int
dump.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = 'cuda:0'
M = 16384
out = []
for name, N, K in (('qkv', 12288, 4096), ('o', 4096, 4096), ('gate_up', 22016, 4096), ('down', 4096, 11008)):
    for lay in ('nt', 'nn', 'tn'):
        if lay == 'nt':
            a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16(); kw = {}
            nt = K // 64
        elif lay == 'nn':   # dX = dY[M,N] @ W[N,K]: contraction N
            a = torch.randn(M, N, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16(); kw = dict(b_n=True)
            nt = N // 64
        else:               # dW = dY^T[N,M] @ X[M,K]: contraction M
            a = torch.randn(M, N, device=dev).bfloat16(); b = torch.randn(M, K, device=dev).bfloat16(); kw = dict(a_t=True, b_n=True)
            nt = M // 64
        for _ in range(3):
            c = ops.gemm(a, b, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c = ops.gemm(a, b, **kw); e1.record(); torch.cuda.synchronize()
        tiles = (c.shape[0] // 256) * (c.shape[1] // 256)
        buf = np.zeros(8 * 16384, dtype=np.uint64)
        assert dump(buf.ctypes.data, buf.size) == 0
        t = buf.reshape(-1, 8)[:min(tiles, 16384)].astype(np.int64)
        clk, rt = t[:, :4], t[:, 4:]
        d_clk = clk[:, 3] - clk[:, 0]
        d_rt = rt[:, 3] - rt[:, 0]
        mhz = (d_clk.sum() / d_rt.sum()) * 100.0
        span_us = (rt[:, 3].max() - rt[:, 0].min()) / 100.0
        r = {'shape': f'{name}.{lay}', 'tiles': tiles, 'ktiles': nt, 'event_us': e0.elapsed_time(e1) * 1e3, 'span_us': float(span_us),
             'shader_mhz': float(mhz), 'cyc_per_ktile': float(np.median((clk[:, 2] - clk[:, 1]) / nt)),
             'cyc_prologue': float(np.median(clk[:, 1] - clk[:, 0])), 'cyc_epilogue': float(np.median(clk[:, 3] - clk[:, 2])),
             'us_per_tile': float(np.median(d_rt) / 100.0), 'tflops': 2.0 * M * N * K / (e0.elapsed_time(e1) * 1e-3) / 1e12,
             'waves_of_tiles': tiles / 256.0}
        # start-time structure: how many distinct dispatch rounds
        starts = np.sort(rt[:, 0] - rt[:, 0].min()) / 100.0
        r['start_us_quartiles'] = [float(np.percentile(starts, q)) for q in (0, 25, 50, 75, 100)]
        out.append(r)
        print(json.dumps(r), flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/gemm4_timing.json', 'w'), indent=1)
