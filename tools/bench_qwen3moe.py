"""Secondary datapoint (NOT the bench.py headline): native DPO step on the Qwen3-30B-A3B LAYER geometry (BASELINE configs[4]
backbone: hidden 2048, 32 heads / 4 kv of 128, 128 experts x 768, top-8, V = 151936) at a depth that fits one GPU with full
AdamW state (the 48-layer model needs expert parallelism), bf16, T = 2048, synthetic data, random init."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align_anything_amd import configs  # noqa: E402
from align_anything_amd.trainers.dpo import DPOTrainer  # noqa: E402
from bench import random_init_  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', type=int, default=2)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--layers', type=int, default=12)
    ap.add_argument('--seq-len', type=int, default=2048)
    ap.add_argument('--response-len', type=int, default=512)
    ap.add_argument('--share-prompt', action='store_true', help="train_cfgs.share_prompt_prefix: the pair's common prefix once per model")
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    cfg = configs.qwen3moe_cfg(2048, 768, a.layers, 32, 4, 151936, 128, 8, head_dim=128)
    B, T, R = a.pairs, a.seq_len, a.response_len
    cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-6, 'lr_warmup_ratio': 0.03, 'weight_decay': 0.0, 'total_training_steps': a.steps + a.warmup, 'share_prompt_prefix': a.share_prompt},
            'model_cfgs': {'pad_token_id': 0}}
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, device=dev)
    random_init_(tr.policy, seed=42)
    tr.reference.load_state_dict(tr.policy.state_dict())
    for g in tr.policy.store.master:
        tr.policy.store.master[g].copy_(tr.policy.store.flat[g])

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        ids = torch.randint(3, 151000, (2 * B, T), generator=g)
        ids[B:, :T - R] = ids[:B, :T - R]
        return {'input_ids': ids.to(dev), 'attention_mask': torch.ones(2 * B, T, dtype=torch.long, device=dev), 'meta_info': {'response_lens': [R] * (2 * B), 'shared_prefix_lens': [T - R] * B}}

    bs = [batch(1), batch(2)]
    for i in range(a.warmup):
        tr.train_step(bs[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = [round(tr.train_step(bs[i % 2])['train/loss'], 5) for i in range(a.steps)]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    h, F, L, V, H, Hkv, hd, k = 2048, 768, a.layers, 151936, 32, 4, 128, 8
    act_gemm = 2.0 * T * (L * (h * (H + 2 * Hkv) * hd + H * hd * h + k * 3 * h * F + h * 128) + h * V)      # ACTIVE params only (top-8)
    attn = L * 2.0 * T * T * H * hd
    per_pair = 8 * (act_gemm + attn)
    print(json.dumps({'workload': f'Qwen3-30B-A3B layer geometry, {L} of 48 layers, DPO step, bf16, T={T}, R={R}, {B} pairs/step',
                      'share_prompt_prefix': a.share_prompt, 'pairs_per_s': B / dt, 'ms_per_step': dt * 1e3, 'ms_per_step_per_layer': dt * 1e3 / L,
                      'active_tflop_per_pair': per_pair / 1e12, 'active_frac_of_dense_bf16_peak': per_pair * B / dt / 2.5e15,
                      'losses': losses, 'params': tr.policy.store.num_params()}))


if __name__ == '__main__':
    main()
