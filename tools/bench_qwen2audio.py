"""Secondary datapoint (NOT the bench.py headline): one native DPO step on the Qwen2-Audio-7B geometry (BASELINE configs[3]
backbone), bf16, T = 2048 with one 30 s clip per pair (128 x 3000 mel frames -> 750 audio tokens), audio tower TRAINABLE (the
reference's default, trainers/text_audio_to_text/dpo.py), synthetic data, random init.  Prints pairs/s and the algorithmic MFMA
fraction (decoder GEMMs + attention, and with the encoder's flops added)."""
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
    ap.add_argument('--layers', type=int, default=32)
    ap.add_argument('--audio-layers', type=int, default=32)
    ap.add_argument('--seq-len', type=int, default=2048)
    ap.add_argument('--response-len', type=int, default=512)
    ap.add_argument('--share-prompt', action='store_true', help='train_cfgs.share_prompt_prefix: the pair\'s common prefix (and its clip) once per model')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    cfg = configs.qwen2_audio_7b(a.layers, a.audio_layers)
    B, T, R = a.pairs, a.seq_len, a.response_len
    cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-6, 'lr_warmup_ratio': 0.03, 'weight_decay': 0.0, 'total_training_steps': a.steps + a.warmup, 'share_prompt_prefix': a.share_prompt},
            'model_cfgs': {'pad_token_id': cfg['pad_token_id']}}
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, device=dev)
    random_init_(tr.policy, seed=42)
    tr.reference.load_state_dict(tr.policy.state_dict())
    for g in tr.policy.store.master:
        tr.policy.store.master[g].copy_(tr.policy.store.flat[g])
    ntok = 750

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        ids = torch.randint(3, 151000, (2 * B, T), generator=g)
        ids[:, 1:1 + ntok] = cfg['audio_token_id']
        ids[B:, :T - R] = ids[:B, :T - R]
        feat = torch.randn(B, 128, 3000, generator=g)
        return {'input_ids': ids.to(dev), 'attention_mask': torch.ones(2 * B, T, dtype=torch.long, device=dev),
                'input_features': torch.cat([feat, feat], 0).to(dev), 'feature_attention_mask': torch.ones(2 * B, 3000, dtype=torch.long, device=dev),
                'meta_info': {'response_lens': [R] * (2 * B), 'shared_prefix_lens': [T - R] * B}}

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
    t, au = cfg['text'], cfg['audio']
    h, F, L, V, H, Hkv, hd = t['hidden_size'], t['intermediate_size'], t['num_layers'], t['vocab_size'], t['num_heads'], t['num_kv_heads'], t['head_dim']
    gemm = 2.0 * T * (L * (h * (H + 2 * Hkv) * hd + H * hd * h + 3 * h * F) + h * V)
    attn = L * 2.0 * T * T * H * hd
    llm_pair = 8 * (gemm + attn)          # policy fwd 2 rows + ref fwd 2 rows + policy bwd (2x)
    d, Fa, La = au['d_model'], au['ffn_dim'], au['num_layers']
    enc_row = 2.0 * 1500 * La * (4 * d * d + 2 * d * Fa) + La * 4.0 * 1500 * 1500 * d        # bidirectional attention: full T^2
    enc_pair = 8 * enc_row                # same 2 + 2 + 2x2 passes: the tower trains
    print(json.dumps({'workload': f'Qwen2-Audio-7B geometry DPO step, bf16, T={T}, R={R}, {B} pairs/step, one 30 s clip (750 audio tokens) per pair, audio tower trainable'
                                  + ('' if (a.layers, a.audio_layers) == (32, 32) else f' [REDUCED DEPTH {a.layers}/{a.audio_layers}]'),
                      'share_prompt_prefix': a.share_prompt, 'pairs_per_s': B / dt, 'ms_per_step': dt * 1e3, 'llm_tflop_per_pair': llm_pair / 1e12, 'encoder_tflop_per_pair': enc_pair / 1e12,
                      'llm_frac_of_dense_bf16_peak': llm_pair * B / dt / 2.5e15, 'total_frac_of_dense_bf16_peak': (llm_pair + enc_pair) * B / dt / 2.5e15,
                      'losses': losses, 'trainable_params': tr.policy.store.num_trainable()}))


if __name__ == '__main__':
    main()
