"""What the asynchronous AdamW really costs the headline step, and whether giving it only PART of the chip helps (round 5).

gemm4 needs a whole compute unit (four waves of 512 registers: nothing else fits on their SIMDs), AdamW is one launch of long-lived grid-stride workgroups on EVERY
compute unit for ~32 ms: while it runs, no gemm4 workgroup of the next step's reference forward can start -- only small-register kernels (the CLIP tower) co-run.  The
engine's side-stream AdamW therefore overlaps with little.  Variants, alternating on one box (bench.py's trainer, LLaVA-1.5-7B geometry, 32 layers):

  async            the shipped engine: AdamW on a side stream over all compute units
  sync             AdamW on the main stream (no overlap at all): async - sync = what the overlap hides today
  masked C         AdamW on a side stream restricted to C compute units (hipExtStreamCreateWithCUMask): the GEMMs keep 256 - C CUs for themselves during a
                   longer update window

-> gpurun_out/r05_adam_window.json"""
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


def stream_on(n_cus, keep):
    """A torch stream that may use `keep` compute units, spread evenly over the groups of 32."""
    words = (n_cus + 31) // 32
    per = [keep // words + (1 if i < keep % words else 0) for i in range(words)]
    mask = [(1 << k) - 1 if k < 32 else 0xFFFFFFFF for k in per]
    arr = (ctypes.c_uint * words)(*mask)
    out = ctypes.c_void_p()
    ops.call('aa_stream_create_cu_mask', arr, words, ctypes.byref(out))
    return torch.cuda.ExternalStream(out.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--pairs', type=int, nargs='*', default=[4, 1])
    ap.add_argument('--cus', type=int, nargs='*', default=[64, 96, 128, 192])
    a = ap.parse_args()
    from align_anything_amd.trainers.dpo import DPOTrainer
    device = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    cfg = configs.llava_1_5_7b(num_layers=32)
    T, R = 2048, 512
    cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-6, 'lr_warmup_ratio': 0.03, 'weight_decay': 0.0, 'adam_betas': [0.9, 0.95], 'lr_scheduler_type': 'cosine',
                           'total_training_steps': 4096, 'freeze_mm_proj': False, 'freeze_language_model': False, 'freeze_vision_tower': True},
            'model_cfgs': {'pad_token_id': cfg['pad_token_id']}}
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, device=device)
    random_init_(tr.policy, seed=42)
    tr.reference.load_state_dict(tr.policy.state_dict())
    for g in tr.policy.store.master:
        tr.policy.store.master[g].copy_(tr.policy.store.flat[g])
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    default_stream = torch.cuda.Stream()
    res = []

    def run(label, B, batches):
        tr.train_step(batches[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            tr.train_step(batches[1 + i])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        r = {'variant': label, 'pairs': B, 'ms_per_step': ms, 'pairs_per_s': B / ms * 1e3}
        print(json.dumps(r), flush=True)
        res.append(r)

    for B in a.pairs:
        batches = [make_batch(cfg, B, T, R, device, seed=4321 + i) for i in range(a.steps + 1)]
        for rep in range(2):
            tr.model.async_optimizer, tr.model._opt_stream = True, default_stream
            run('async', B, batches)
            tr.model.async_optimizer = False
            run('sync', B, batches)
            for C in a.cus:
                tr.model.wait_optimizer()
                torch.cuda.synchronize()
                tr.model.async_optimizer, tr.model._opt_stream = True, stream_on(n_cus, C)
                run(f'masked {C}', B, batches)
        tr.model.wait_optimizer()
        torch.cuda.synchronize()
        del batches
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'r05_adam_window.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
