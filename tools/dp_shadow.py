"""A one-GPU MODEL of what data-parallel gradient reduction costs the headline step -- NOT a scaling curve, and labelled as a model
wherever it is quoted.  DESIGN.md section 6 argues on paper that RCCL, resident on C compute units during backward, pushes every 1024-tile gemm4 launch
from 4 rounds of 256 workgroups to 5 (+25 % on those launches).  Here that arithmetic is measured on the real step (bench.py's trainer: LLaVA-1.5-7B
geometry, 4 pairs, T = 2048, 32 layers):

  mask C     the backward pass runs on a stream whose kernels may use only 256 - C compute units (hipExtStreamCreateWithCUMask: the CUs a collective
             would hold), C in {8, 16, 32}; the forward passes and the optimizer keep all 256
  traffic C  a kernel of C long-lived workgroups streams the 13.5 GB of bf16 gradients src -> dst (read + write) on a side stream for the length of the
             backward pass: the collective's HBM traffic, issued from CUs the GEMMs then cannot use (a workgroup of gemm4 needs a whole CU)
  both       mask C on the compute stream AND the traffic kernel on the masked-out CUs' side

Per variant: step time and backward time (HIP events), against the unmodified step on the same box, alternating.  -> gpurun_out/r05_dp_shadow.json

What it does not model: link latency / bandwidth (the exchange itself), RCCL's proxy threads, the all-reduce's dependence on the bucket order.  It answers
one question: what does the compute stream lose when C CUs and ~X GB/s of HBM belong to something else during backward."""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from align_anything_amd import configs, ops  # noqa: E402
from bench import make_batch, random_init_  # noqa: E402


def masked_stream(n_cus, take):
    """A torch stream restricted to n_cus - take compute units.  The mask clears the LAST `take` bits of every group of 32 CUs evenly (8 XCDs x 32 CUs: a
    collective's workgroups are spread by the dispatcher, one XCD does not lose them all)."""
    words = (n_cus + 31) // 32
    mask = [0xFFFFFFFF] * words
    per = [take // words + (1 if i < take % words else 0) for i in range(words)]
    for i, k in enumerate(per):
        for b in range(k):
            mask[i] &= ~(1 << (31 - b))
    arr = (ctypes.c_uint * words)(*mask)
    out = ctypes.c_void_p()
    ops.call('aa_stream_create_cu_mask', arr, words, ctypes.byref(out))
    return torch.cuda.ExternalStream(out.value), [hex(m) for m in mask]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--pairs', type=int, default=4)
    ap.add_argument('--layers', type=int, default=32)
    ap.add_argument('--cus', type=int, nargs='*', default=[8, 16, 32])
    ap.add_argument('--out', default='r05_dp_shadow.json')
    a = ap.parse_args()
    from align_anything_amd.trainers.dpo import DPOTrainer
    device = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    cfg = configs.llava_1_5_7b(num_layers=a.layers)
    B, T, R = a.pairs, 2048, 512
    cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-6, 'lr_warmup_ratio': 0.03, 'weight_decay': 0.0, 'adam_betas': [0.9, 0.95], 'lr_scheduler_type': 'cosine',
                           'total_training_steps': 4096, 'freeze_mm_proj': False, 'freeze_language_model': False, 'freeze_vision_tower': True},
            'model_cfgs': {'pad_token_id': cfg['pad_token_id']}}
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, device=device)
    random_init_(tr.policy, seed=42)
    tr.reference.load_state_dict(tr.policy.state_dict())
    for g in tr.policy.store.master:
        tr.policy.store.master[g].copy_(tr.policy.store.flat[g])
    batches = [make_batch(cfg, B, T, R, device, seed=1234 + i) for i in range(a.steps + 1)]
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    grad_bytes = sum(t.numel() * t.element_size() for t in tr.policy.store.gflat.values())
    src = torch.empty(1 << 30, dtype=torch.uint8, device=device)          # 1 GiB windows, walked `passes` times: the bucket stream of a step
    dst = torch.empty(1 << 30, dtype=torch.uint8, device=device)
    side = torch.cuda.Stream()
    orig_backward = tr.model.backward
    state = {'mask': None, 'traffic': 0, 'bwd': []}

    def backward(loss=None):
        cur = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        if state['traffic']:
            side.wait_event(e0)
            with torch.cuda.stream(side):
                passes = max(1, round(grad_bytes / src.numel()))
                ops.call('aa_shadow_traffic', src.data_ptr(), dst.data_ptr(), src.numel(), int(state['traffic']), int(passes), ops.stream())
        if state['mask'] is not None:
            ms = state['mask']
            ms.wait_event(e0)
            with torch.cuda.stream(ms):
                orig_backward(loss)
                done = torch.cuda.Event()
                done.record(ms)
            cur.wait_event(done)
        else:
            orig_backward(loss)
        e1.record(cur)
        state['bwd'].append((e0, e1))

    tr.model.backward = backward

    def run(label, mask=None, traffic=0):
        state.update(mask=mask, traffic=traffic, bwd=[])
        tr.train_step(batches[0])
        torch.cuda.synchronize()
        state['bwd'] = []
        t0 = time.perf_counter()
        for i in range(a.steps):
            tr.train_step(batches[1 + i])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps * 1e3
        bwd = sum(x.elapsed_time(y) for x, y in state['bwd']) / len(state['bwd'])
        if traffic:
            side.synchronize()
        r = {'variant': label, 'ms_per_step': dt, 'backward_ms': bwd}
        print(json.dumps(r), flush=True)
        return r

    res = {'what': 'one-GPU MODEL of a resident collective beside the backward pass (tools/dp_shadow.py) -- not a multi-GPU measurement',
           'workload': f'bench.py trainer, {B} pairs, T {T}, {a.layers} layers', 'compute_units': n_cus, 'gradient_bytes_per_step': grad_bytes, 'runs': []}
    res['runs'].append(run('baseline'))
    for C in a.cus:
        ms, mask = masked_stream(n_cus, C)
        r = run(f'mask {C}', mask=ms)
        r['cu_mask'] = mask
        res['runs'].append(r)
        res['runs'].append(run(f'traffic {C}', traffic=C))
        res['runs'].append(run(f'both {C}', mask=ms, traffic=C))
        res['runs'].append(run('baseline'))
    base = [r for r in res['runs'] if r['variant'] == 'baseline']
    b_step, b_bwd = sum(r['ms_per_step'] for r in base) / len(base), sum(r['backward_ms'] for r in base) / len(base)
    for r in res['runs']:
        r['step_vs_baseline'] = r['ms_per_step'] / b_step
        r['backward_vs_baseline'] = r['backward_ms'] / b_bwd
    res['baseline_mean'] = {'ms_per_step': b_step, 'backward_ms': b_bwd}
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', a.out), 'w'), indent=1)


if __name__ == '__main__':
    main()
