"""kind = "reference" CPU baseline: the REFERENCE'S OWN trainer code --
align_anything/trainers/text_image_to_text/dpo.py:107-166 `DPOTrainer.loss` (compute_log_probs of policy and reference on HF
LlavaForConditionalGeneration) + backward + global-norm clip + AdamW, i.e. the body of `train_step`
(trainers/text_to_text/dpo.py:205-237) -- timed on the build container's host cores on a DEPTH-REDUCED LLaVA-1.5-7B (full width:
h = 4096, 32 heads, ffn 11008, V = 32064, CLIP-L/14-336 tower with 24 layers, 576 image tokens, T = 2048, one pair), fp32.

The reference cannot travel to the GPU box (/root/reference is absent there), so this runs HERE, once, through oracle/_shim.py
(unmodified reference modules; DeepSpeed's engine is replaced by loss.backward() + torch.optim.AdamW, its FusedAdam being absent),
and the result is committed as profiles/cpu_reference_L2.json; bench.py attaches it to `cpu_baseline.reference`.  Depth L in {1, 2}
decoder layers is measured and the per-pair time at L = 32 extrapolated linearly (t = fixed + L x per_layer; fixed = embedding,
lm_head over all T positions as the reference executes it, CLIP tower x 4 passes, projector).

    python tools/cpu_reference_baseline.py [--threads N] [--reps 2]        (TEST / MEASUREMENT INFRASTRUCTURE, not product code)
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(L):
    import torch
    from transformers import CLIPVisionConfig, LlamaConfig, LlavaConfig, LlavaForConditionalGeneration
    vc = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336, patch_size=14)
    tc = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=L, num_attention_heads=32, num_key_value_heads=32,
                     vocab_size=32064, rms_norm_eps=1e-5, max_position_embeddings=4096)
    cfg = LlavaConfig(vision_config=vc, text_config=tc, image_token_id=32000, image_seq_length=576)
    torch.manual_seed(0)
    return LlavaForConditionalGeneration(cfg)


def time_depth(L, reps, T=2048, R=512):
    import torch
    from align_anything.trainers.text_image_to_text.dpo import DPOTrainer
    from align_anything.utils.tools import dict_to_namedtuple
    policy, refm = build(L).train(), build(L).eval()
    for p in refm.parameters():
        p.requires_grad_(False)
    # configs/train/text_image_to_text/dpo.yaml:62-66: vision tower frozen, projector + language model trained
    for n, p in policy.named_parameters():
        if 'vision_tower' in n:
            p.requires_grad_(False)
    opt = torch.optim.AdamW([p for p in policy.parameters() if p.requires_grad], lr=1e-6, betas=(0.9, 0.95), weight_decay=0.0)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(3, 32000, (2, T), generator=g)
    ids[:, 0] = 1
    ids[:, 1:577] = 32000
    ids[1, :T - R] = ids[0, :T - R]
    pix = torch.randn(1, 3, 336, 336, generator=g)
    batch = {'input_ids': ids, 'attention_mask': torch.ones(2, T, dtype=torch.long), 'pixel_values': torch.cat([pix, pix], 0),
             'meta_info': {'response_lens': [R, R]}}
    tr = DPOTrainer.__new__(DPOTrainer)
    tr.cfgs = dict_to_namedtuple({'train_cfgs': {'scale_coeff': 0.1}})
    tr.tokenizer = SimpleNamespace(pad_token_id=32001)
    tr.infer_batch = lambda b: {k: v for k, v in b.items() if k != 'meta_info'}
    tr.model, tr.reference_model = SimpleNamespace(module=policy), SimpleNamespace(module=refm)

    def step():
        t0 = time.time()
        ld = tr.loss(batch)                      # the reference's own loss(): policy + reference compute_log_probs, sigmoid-logratio
        ld['loss'].backward()                    # DeepSpeedEngine.backward
        torch.nn.utils.clip_grad_norm_([p for p in policy.parameters() if p.requires_grad], 1.0)      # ds_cfgs gradient_clipping 1.0
        opt.step()
        opt.zero_grad(set_to_none=True)
        return time.time() - t0, float(ld['loss'])
    warm, _ = step()
    times = sorted(step()[0] for _ in range(reps))
    return {'layers': L, 'warmup_s': round(warm, 2), 'timed_s': [round(t, 2) for t in times], 'median_s': times[len(times) // 2]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--reps', type=int, default=2)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'cpu_reference_L2.json'))
    a = ap.parse_args()
    import torch
    from oracle import _shim
    _shim.install()
    if a.threads:
        torch.set_num_threads(a.threads)
    cores = torch.get_num_threads()
    runs = [time_depth(L, a.reps) for L in (1, 2)]
    per_layer = runs[1]['median_s'] - runs[0]['median_s']
    fixed = runs[0]['median_s'] - per_layer
    t32 = fixed + 32 * per_layer
    import platform
    cpu = ''
    try:
        cpu = [l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')][0]
    except (OSError, IndexError):
        pass
    out = {'kind': 'reference', 'value': 1.0 / t32, 'unit': 'pairs/s', 'cores': cores, 'cpu': cpu or platform.processor(),
           'where': 'build container (the reference is not on the GPU box)', 'dtype': 'f32',
           'what': "align_anything DPOTrainer.loss (text_image_to_text/dpo.py:107-166) on HF LlavaForConditionalGeneration + backward + "
                   'clip_grad_norm_ + torch.optim.AdamW (DeepSpeed absent), one pair, T = 2048, 576 image tokens, response 512',
           'sample': f'full-width LLaVA-1.5-7B at depth L = 1 and L = 2 (CLIP-L tower 24 layers, lm_head over all T, as the reference executes), '
                     f'{a.reps} timed steps each after a warm-up; per-pair seconds at L = 32 extrapolated: fixed {fixed:.1f} s + 32 x {per_layer:.1f} s = {t32:.1f} s',
           'runs': runs, 'seconds_per_pair_L32_extrapolated': t32, 'per_layer_s': per_layer, 'fixed_s': fixed,
           'script': 'tools/cpu_reference_baseline.py'}
    with open(a.out, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
