"""Secondary datapoint (NOT the bench.py headline): one native DPO step on the Qwen2-VL-7B geometry (BASELINE configs[2]
backbone), bf16, T = 2048 with one 448x448 image per pair (1024 patches -> 256 image tokens), synthetic data, random
init.  Prints pairs/s and the algorithmic MFMA fraction of the LLM GEMMs + attention."""
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
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--layers', type=int, default=28)
    ap.add_argument('--vision-depth', type=int, default=32)
    ap.add_argument('--seq-len', type=int, default=2048)
    ap.add_argument('--response-len', type=int, default=512)
    ap.add_argument('--share-prompt', action='store_true', help="train_cfgs.share_prompt_prefix: the pair's common prefix (and its image) once per model")
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    cfg = configs.qwen2_vl_7b(a.layers, a.vision_depth)
    B, T, R = a.pairs, a.seq_len, a.response_len
    cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-6, 'lr_warmup_ratio': 0.03, 'weight_decay': 0.0, 'total_training_steps': a.steps + a.warmup, 'share_prompt_prefix': a.share_prompt},
            'model_cfgs': {'pad_token_id': cfg['pad_token_id']}}
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, device=dev, share_vision_tower=False)
    random_init_(tr.policy, seed=42)
    tr.policy.vision.invalidate()
    tr.reference.load_state_dict(tr.policy.state_dict())
    for g in tr.policy.store.master:
        tr.policy.store.master[g].copy_(tr.policy.store.flat[g])
    grid = [[1, 32, 32]] * (2 * B)
    ntok = 256

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        ids = torch.randint(3, 151000, (2 * B, T), generator=g)
        ids[:, 0] = 1
        ids[:, 1:1 + ntok] = cfg['image_token_id']
        ids[B:, :T - R] = ids[:B, :T - R]
        pix = torch.randn(B * 1024, 1176, generator=g)
        return {'input_ids': ids.to(dev), 'attention_mask': torch.ones(2 * B, T, dtype=torch.long, device=dev),
                'pixel_values': torch.cat([pix, pix], 0).to(dev), 'image_grid_thw': grid, 'meta_info': {'response_lens': [R] * (2 * B), 'shared_prefix_lens': [T - R] * B}}

    bs = [batch(1), batch(2)]
    for i in range(a.warmup):
        tr.train_step(bs[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = []
    for i in range(a.steps):
        losses.append(round(tr.train_step(bs[i % 2])['train/loss'], 5))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    t = cfg['text']
    h, F, L, V, H, Hkv, hd = t['hidden_size'], t['intermediate_size'], t['num_layers'], t['vocab_size'], t['num_heads'], t['num_kv_heads'], t['head_dim']
    gemm = 2.0 * T * (L * (h * (H + 2 * Hkv) * hd + H * hd * h + 3 * h * F) + h * V)
    attn = L * 2.0 * T * T * H * hd
    per_pair = 8 * (gemm + attn)          # policy fwd 2 rows + ref fwd 2 rows + policy bwd (2x) ; vision excluded
    print(json.dumps({'workload': f'Qwen2-VL-7B geometry DPO step, bf16, T={T}, R={R}, {B} pairs/step, 1 image (1024 patches) per pair'
                                  + ('' if a.layers == 28 else f' [REDUCED DEPTH {a.layers}/{a.vision_depth}]'),
                      'share_prompt_prefix': a.share_prompt, 'pairs_per_s': B / dt, 'ms_per_step': dt * 1e3, 'llm_tflop_per_pair': per_pair / 1e12,
                      'llm_frac_of_dense_bf16_peak': per_pair * B / dt / 2.5e15, 'losses': losses,
                      'trainable_params': tr.policy.store.num_trainable()}))


if __name__ == '__main__':
    main()
