"""Timing build (see gemm4_timing.py): cycles per K-tile as a function of how loaded the memory system is -- few tiles (all operands L2 / MALL
resident, 16..64 CUs busy) vs one full round vs many rounds, long K loops."""
import ctypes, json, os, sys
import numpy as np
import torch
from align_anything_amd import ops
from align_anything_amd.lib import LIB

dll = LIB.load()
dump = dll.aa_gemm4_timing_dump
dump.restype = ctypes.c_int
dump.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = 'cuda:0'
ops.gemm_set_tile(5)          # force the 256x256 4-wave kernel also where the heuristic prefers smaller tiles
out = []
for M, N, K in ((2048, 2048, 16384), (4096, 4096, 16384), (16384, 4096, 16384), (4096, 4096, 4096), (16384, 12288, 4096), (16384, 22016, 4096), (16384, 4096, 11008)):
    for lay in os.environ.get('G4T_LAYOUTS', 'nt,nn,tn').split(','):
        if lay == 'nt':
            a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16(); kw = {}
        elif lay == 'nn':
            a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(K, N, device=dev).bfloat16(); kw = dict(b_n=True)
        else:
            a = torch.randn(K, M, device=dev).bfloat16(); b = torch.randn(K, N, device=dev).bfloat16(); kw = dict(a_t=True, b_n=True)
        nt = K // 64
        for _ in range(6):
            c = ops.gemm(a, b, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c = ops.gemm(a, b, **kw); e1.record(); torch.cuda.synchronize()
        tiles = (M // 256) * (N // 256)
        buf = np.zeros(8 * 16384, dtype=np.uint64)
        assert dump(buf.ctypes.data, buf.size) == 0
        t = buf.reshape(-1, 8)[:min(tiles, 16384)].astype(np.int64)
        clk, rt = t[:, :4], t[:, 4:]
        mhz = ((clk[:, 3] - clk[:, 0]).sum() / (rt[:, 3] - rt[:, 0]).sum()) * 100.0
        r = {'M': M, 'N': N, 'K': K, 'lay': lay, 'tiles': tiles, 'shader_mhz': round(float(mhz)), 'cyc_per_ktile': round(float(np.median((clk[:, 2] - clk[:, 1]) / nt)), 1),
             'cyc_per_ktile_min': round(float(((clk[:, 2] - clk[:, 1]) / nt).min()), 1), 'wait_cyc_per_ktile': round(float(np.median(t[:, 5] / nt)), 1), 'barrier_cyc_per_ktile': round(float(np.median(t[:, 6] / nt)), 1),               'cyc_epilogue': float(np.median(clk[:, 3] - clk[:, 2])), 'tflops': round(2.0 * M * N * K / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)}
        out.append(r)
        print(json.dumps(r), flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/gemm4_timing2.json', 'w'), indent=1)
