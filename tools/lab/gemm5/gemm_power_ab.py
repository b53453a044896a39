"""Why the 32x32x16 kernel (tools/lab/gemm5) loses although it needs fewer cycles: run each variant back-to-back for a few seconds on one
hot shape while sampling the package power and the shader clock (rocm-smi), and report TFLOP/s, average sclk and watts per variant.
    python tools/gemm_power_ab.py  ->  gpurun_out/r03_gemm_power_ab.json"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align_anything_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
M, N, K = 16384, 4096, 11008
SECONDS = float(os.environ.get('AA_POWER_SECONDS', 4))


def smi_sample():
    try:
        r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=5)
        d = json.loads(r.stdout)
        card = d[sorted(d)[0]]
        out = {}
        for k, v in card.items():
            kl = k.lower()
            if 'sclk' in kl and 'clock' in kl:
                out['sclk_mhz'] = float(str(v).strip('()').lower().replace('mhz', ''))
            if 'power' in kl and ('socket' in kl or 'average' in kl or 'current' in kl):
                try:
                    out['power_w'] = float(v)
                except ValueError:
                    pass
        return out or {'raw': card}
    except Exception as ex:      # the line must still be produced
        return {'error': repr(ex)}


def run(layout, m32):
    a_t, b_n = layout == 'tn', layout in ('nn', 'tn')
    if layout == 'nn':
        m, n, k = M, K, N
    elif layout == 'tn':
        m, n, k = K, N, M
    else:
        m, n, k = M, N, K
    rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
    a = rnd(k, m) if a_t else rnd(m, k)
    b = rnd(k, n) if b_n else rnd(n, k)
    out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    ops.gemm_set_tile(5)
    ops.gemm_set_mfma32(m32)
    for _ in range(5):
        ops.gemm(a, b, out=out, a_t=a_t, b_n=b_n)
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            samples.append(smi_sample())
            time.sleep(0.25)
    th = threading.Thread(target=poll)
    th.start()
    n_launch, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(50):
            ops.gemm(a, b, out=out, a_t=a_t, b_n=b_n)
        torch.cuda.synchronize()
        n_launch += 50
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    sk = [s['sclk_mhz'] for s in samples[2:] if 'sclk_mhz' in s]
    pw = [s['power_w'] for s in samples[2:] if 'power_w' in s]
    return {'layout': layout, 'kernel': 'gemm5 (32x32x16)' if m32 else 'gemm4 (16x16x32)', 'shape': [m, n, k], 'tflops': 2.0 * m * n * k * n_launch / dt / 1e12,
            'sclk_mhz_avg': sum(sk) / len(sk) if sk else None, 'power_w_avg': sum(pw) / len(pw) if pw else None, 'smi_samples': len(samples),
            'first_sample': samples[0] if samples else None}


res = []
for layout in ('nt', 'nn', 'tn'):
    for rep in range(2):
        for m32 in (False, True):
            r = run(layout, m32)
            r['rep'] = rep
            print(r, flush=True)
            res.append(r)
ops.gemm_set_tile(-1)
ops.gemm_set_mfma32(False)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/r03_gemm_power_ab.json', 'w'), indent=1)
