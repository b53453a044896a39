"""Build libaa_hip.so (the C-ABI boundary, include/aa_hip.h) in-tree with hipcc for gfx950.

No torch dependency: the library takes raw device pointers + a hipStream_t.  Objects are cached by
source mtime so repeated builds are cheap.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, 'build')
LIB = os.path.join(HERE, 'libaa_hip.so')
SOURCES = ['runtime.hip', 'comm.hip', 'rl_math.hip', 'lmhead.hip', 'pref_losses.hip', 'elementwise.hip', 'elementwise_f32.hip', 'optim.hip', 'gemm.hip', 'gemm4.hip', 'attention.hip', 'decode.hip',
           'gemm_f32.hip', 'attention_f32.hip', 'moe.hip', 'moe_f32.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast', '-Wno-unused-result']
FLAGS += os.environ.get('AA_HIPCC_EXTRA', '').split()      # experiment builds only (e.g. -DAA_G4_TIMING, tools/gemm4_timing.py)


def _newer(a: str, b: str) -> bool:
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _compile(src: str) -> str:
    s = os.path.join(CSRC, src)
    o = os.path.join(OBJ, src.replace('.hip', '.o'))
    hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(('.h', '.inc'))]      # every header / generated include (gemm4_*.inc)
    if src.endswith('_f32.hip') and os.path.exists(os.path.join(CSRC, src.replace('_f32.hip', '.hip'))):
        hdrs.append(os.path.join(CSRC, src.replace('_f32.hip', '.hip')))   # twin instantiation includes the bf16 source
    if _newer(s, o) or any(_newer(h, o) for h in hdrs if os.path.exists(h)):
        cmd = [HIPCC, *FLAGS, '-c', s, '-o', o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
    return o


def build(verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    if any(_newer(o, LIB) for o in objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-ldl', '-o', LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(verbose=True)
    sys.exit(0)
