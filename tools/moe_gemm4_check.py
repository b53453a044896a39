"""Check + time the row-grouped MoE GEMMs on the gemm4 tile (csrc/gemm4.hip GRP, shipped in round 4; AA_MOE_GEMM4=1 -> ops.MOE_ALIGN = 256, aa_gemm4_grouped) against the
128 x 256 8-wave kernel (AA_MOE_GEMM4=0) at the Qwen3-30B-A3B layer geometry: 8192 tokens x top-8 of 128 experts, h = 2048, F = 768.

    python tools/moe_gemm4_check.py            parent: one child per setting, prints both JSON lines
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from align_anything_amd import ops
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    rows, E, k, h, F = 8192, 128, 8, 2048, 768
    idx = torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(rows)]).to(torch.int32).to(dev)
    plan = ops.moe_plan(idx, E)
    x = (torch.randn(rows, h, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    xp = ops.moe_gather(x, plan['src'])
    wgu = (torch.randn(E, 2 * F, h, generator=g) * 0.02).to(torch.bfloat16).to(dev)
    wdn = (torch.randn(E, h, F, generator=g) * 0.02).to(torch.bfloat16).to(dev)
    off = plan['off'].cpu().tolist()
    out = {'AA_MOE_GEMM4': os.environ.get('AA_MOE_GEMM4', '0'), 'align': ops.MOE_ALIGN, 'cap': plan['cap'], 'rows_in_use': off[E]}

    def rel(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))

    gu = ops.gemm_grouped(xp, wgu, plan)                              # forward NT
    act = ops.swiglu_fwd(gu)
    y = ops.gemm_grouped(act, wdn, plan)
    dgu = ops.gemm_grouped(y, wdn, plan, b_n=True)                    # dX NN: [cap, h] x [E, h, F] -> [cap, F]
    torch.cuda.synchronize()
    worst = {'fwd_gu': 0.0, 'fwd_down': 0.0, 'dx': 0.0}
    for e in (0, 1, 17, 63, 127):
        a, b = off[e], off[e + 1]
        if b == a:
            continue
        worst['fwd_gu'] = max(worst['fwd_gu'], rel(gu[a:b].float(), xp[a:b].float() @ wgu[e].float().t()))
        worst['fwd_down'] = max(worst['fwd_down'], rel(y[a:b].float(), act[a:b].float() @ wdn[e].float().t()))
        worst['dx'] = max(worst['dx'], rel(dgu[a:b].float(), y[a:b].float() @ wdn[e].float()))
    out['rel_err_vs_torch'] = worst
    tail = gu[off[E]:]
    out['tail_zero'] = bool(tail.numel() == 0 or float(tail.float().abs().max()) == 0.0)
    pad = plan['src'].cpu() < 0
    out['pad_rows_zero'] = bool(float(gu.cpu()[pad].float().abs().max()) == 0.0) if bool(pad.any()) else True
    for name, fn, fl in (('fwd_gu', lambda: ops.gemm_grouped(xp, wgu, plan, out=gu), 2.0 * rows * k * h * 2 * F),
                         ('fwd_down', lambda: ops.gemm_grouped(act, wdn, plan, out=y), 2.0 * rows * k * F * h),
                         ('dx_down', lambda: ops.gemm_grouped(y, wdn, plan, out=dgu, b_n=True), 2.0 * rows * k * F * h)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        out[name] = {'us': round(ms * 1e3, 1), 'tflops_on_real_rows': round(fl / (ms * 1e-3) / 1e12, 1)}
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child()
    else:
        for v in ('0', '1'):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, AA_MOE_GEMM4=v), capture_output=True, text=True)
            print(r.stdout.strip() or r.stderr[-2000:])
