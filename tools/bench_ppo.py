"""BASELINE configs[2] on ONE MI355X: one full text+image -> text PPO iteration (align_anything/trainers/text_image_to_text/ppo.py:206-379:
`rollout` = generate + reward score + actor / reference log-probs + critic values, then `rl_step` = actor and critic updates) on the
Qwen2-VL-7B geometry with all FOUR models resident -- actor (trainable), reference (frozen), reward model (frozen), critic (trainable) --
the HIP decode rollout replacing vLLM / HF generate.  Not the bench.py headline; VERDICT r2 item 4.

Memory plan (bf16 weights and gradients, fp32 master / m / v = 16 B per trainable parameter, 2 B per frozen one; ViT 0.675 B frozen):
    actor   7.615 B text params x 16 B = 121.8 GB      critic (score head, no lm_head) 7.07 B x 16 B = 113.1 GB
    reference 15.2 GB + reward 14.1 GB frozen, 4 x 1.35 GB towers = 5.4 GB                              => 269.7 GB = 251 GiB persistent
    rollout: strip-major decode copies of the actor (+13.6 GB), KV cache (57 KB per token), prefill activations
    update:  saved activations of ONE response row (~4.2 MB per token) -- the reference's per_device_prompt_batch_size = 1
of the 288 GiB of one MI355X.  Nothing is recomputed or offloaded; the reference needs ZeRO-3 across 8 GPUs for the same four models
(configs/train/text_image_to_text/ppo.yaml:25-35, actor / critic_gradient_checkpointing: True).

    python tools/bench_ppo.py [--prompts 1] [--new-tokens 512] [--iters 2] [--layers 28] [--vision-depth 32]

Prints one JSON line: rollout tokens/s, ms per decode position, prefill ms, scoring ms, rl_step ms, peak HBM, and the split."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align_anything_amd import configs  # noqa: E402
from bench import random_init_  # noqa: E402


def build_trainer(cfg, new_tokens, total_steps, device):
    from align_anything_amd.trainers.ppo_ti2t import PPOTrainerTI2T
    cfgs = {'train_cfgs': {'actor_lr': 1e-5, 'critic_lr': 5e-6, 'actor_weight_decay': 0.01, 'critic_weight_decay': 0.0, 'actor_lr_warmup_ratio': 0.03,
                           'critic_lr_warmup_ratio': 0.03, 'actor_lr_scheduler_type': 'cosine', 'critic_lr_scheduler_type': 'constant', 'adam_betas': [0.9, 0.95],
                           'kl_coeff': 0.02, 'clip_range_ratio': 0.2, 'clip_range_value': 5.0, 'clip_range_score': 50.0, 'gamma': 1.0, 'gae_lambda': 0.95,
                           'total_training_steps': total_steps},
            'model_cfgs': {'pad_token_id': cfg['pad_token_id'], 'max_new_tokens': new_tokens, 'eos_token_id': None, 'temperature': 1.0, 'top_p': 1.0,
                           'repetition_penalty': 1.0, 'model_max_length': 2048}}
    tr = PPOTrainerTI2T(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, device=device)
    random_init_(tr.actor_model.module, seed=42)
    sd = tr.actor_model.module.state_dict()
    tr.actor_reference_model.module.load_state_dict(sd)
    for eng, seed in ((tr.reward_model, 43), (tr.reward_critic_model, 44)):
        random_init_(eng.module, seed=seed)
    del sd
    for eng in (tr.actor_model, tr.reward_critic_model):        # the fp32 masters follow the re-initialised bf16 weights
        st = eng.module.store
        for g in st.master:
            if st.master[g] is not st.flat[g]:
                st.master[g].copy_(st.flat[g])
    for eng in (tr.actor_model, tr.actor_reference_model, tr.reward_model, tr.reward_critic_model):
        if hasattr(eng.module, 'vision'):
            eng.module.vision.invalidate()
    return tr


def prompt_batch(cfg, N, device, seed, text_tokens=60, grid=(1, 32, 32)):
    """What the Qwen2-VL processor + PromptOnlyCollator hand over: <bos> <vision_start> 256 image tokens <vision_end> text, one
    448 x 448 image (1024 patches of 1176 values) per prompt, no padding (equal lengths)."""
    g = torch.Generator().manual_seed(seed)
    ntok = grid[1] * grid[2] // 4
    ids = torch.cat([torch.tensor([[1, 151652]]).expand(N, 2), torch.full((N, ntok), cfg['image_token_id']), torch.full((N, 1), 151653),
                     torch.randint(3, 151000, (N, text_tokens), generator=g)], 1)
    pix = torch.randn(N * grid[0] * grid[1] * grid[2], 1176, generator=g)
    return {'input_ids': ids.to(device), 'attention_mask': torch.ones_like(ids).to(device), 'pixel_values': pix.to(device),
            'image_grid_thw': [list(grid)] * N}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--prompts', type=int, default=1, help='prompts per rollout (reference default per_device_prompt_batch_size: 1)')
    ap.add_argument('--new-tokens', type=int, default=512)
    ap.add_argument('--iters', type=int, default=2)
    ap.add_argument('--layers', type=int, default=28)
    ap.add_argument('--vision-depth', type=int, default=32)
    a = ap.parse_args()
    print(json.dumps(bench(a.prompts, a.new_tokens, a.iters, a.layers, a.vision_depth)), flush=True)


def bench(prompts=1, new_tokens=512, iters=2, layers=28, vision_depth=32, device='cuda:0'):
    dev = torch.device(device)
    cfg = configs.qwen2_vl_7b(layers, vision_depth)
    torch.cuda.set_device(dev)
    torch.zeros(1, device=dev)                                    # create the context before the memory statistics are touched
    torch.cuda.reset_peak_memory_stats(dev)
    t0 = time.perf_counter()
    tr = build_trainer(cfg, new_tokens, total_steps=iters + 2, device=dev)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    resident = torch.cuda.memory_allocated(dev)

    def clock(fn):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return out, (time.perf_counter() - t1) * 1e3

    from align_anything_amd.generation import generate
    rows = []
    prefill_ms = None
    for it in range(iters + 1):                                   # iteration 0 = warm-up (allocations, decode copies, GEMM plans)
        pb = prompt_batch(cfg, prompts, dev, seed=100 + it)
        gen = torch.Generator(device=dev).manual_seed(7 + it)
        (actor_batch, lens), ms_gen = clock(lambda: tr.actor_step(pb, gen))
        # everything else of `rollout` (reward score, actor / reference log-probs, critic values) on the sequences just generated:
        # equal-length prompts and no EOS -> the padding-left form IS the raw generate output rollout(sequences=) expects
        (inf, training), ms_score = clock(lambda: tr.rollout(pb, sequences=actor_batch['input_ids']))
        info, ms_rl = clock(lambda: tr.rl_step(inf, training))
        if it == iters:                                           # prefill alone (one generated token), after the timed iterations
            mm = {'image_grid_thw': pb['image_grid_thw']}
            # inside the same few-row GEMM scope as tr.actor_step (ops.few_row_gemms: split-K prefill), so that generate - prefill is the decode loop alone
            from align_anything_amd import ops
            _, prefill_ms = clock(ops.few_row_gemms(lambda: generate(tr.actor_model.module, pb['input_ids'], pb['attention_mask'], max_new_tokens=1, do_sample=False,
                                                                     pad_token_id=cfg['pad_token_id'], pixel_values=pb['pixel_values'], **mm)))
        rows.append({'iteration': it, 'warmup': it == 0, 'generate_ms': round(ms_gen, 2), 'score_ms': round(ms_score, 2), 'rl_step_ms': round(ms_rl, 2),
                     'response_lens': lens, 'actor_loss': info['train/actor_loss'], 'critic_loss': info['train/reward_critic_loss'],
                     'kl': info['train/kl_divergence'], 'mean_generated_length': info['train/mean_generated_length']})
    timed = rows[1:]
    avg = lambda k: sum(r[k] for r in timed) / len(timed)
    gen_ms, score_ms, rl_ms = avg('generate_ms'), avg('score_ms'), avg('rl_step_ms')
    T0 = pb['input_ids'].shape[1]
    decode_ms = (gen_ms - prefill_ms) / max(1, new_tokens - 1)
    st_a, st_c = tr.actor_model.module.store, tr.reward_critic_model.module.store
    weights_gb = sum(eng.module.store.num_params() for eng in (tr.actor_model, tr.actor_reference_model, tr.reward_model, tr.reward_critic_model)) * 2 / 1e9
    # decode streams every decoder matrix + lm_head once per position
    t = cfg['text']
    stream_bytes = 2.0 * (t['num_layers'] * (t['hidden_size'] * (t['num_heads'] + 2 * t['num_kv_heads']) * t['head_dim'] + t['hidden_size'] * t['num_heads'] * t['head_dim']
                                             + 3 * t['hidden_size'] * t['intermediate_size']) + t['vocab_size'] * t['hidden_size'])
    out = {
        'metric': 'one text+image -> text PPO iteration (rollout + rl_step), Qwen2-VL-7B geometry, 4 models resident, 1x MI355X',
        'config': {'workload': f'BASELINE configs[2] on one GPU: {prompts} prompt(s) x (2 + 256 image tokens + 1 + 60 text = {T0} tokens) -> {new_tokens} new tokens '
                               f'(sampling, no EOS), then rl_step on the {prompts} sequence(s); {layers} decoder layers, ViT depth {vision_depth}'
                               + ('' if (layers, vision_depth) == (28, 32) else ' [REDUCED DEPTH]'),
                   'reference_defaults': 'per_device_prompt_batch_size 1, per_device_train_batch_size 1, max_new_tokens 512 (configs/train/text_image_to_text/ppo.yaml:25-29,155-164)'},
        'dtype': 'bf16', 'data': 'synthetic', 'iters': iters, 'warmup': 1,
        'iteration_ms': gen_ms + score_ms + rl_ms,
        'split_ms': {'generate': gen_ms, 'prefill_of_generate': prefill_ms, 'decode_of_generate': gen_ms - prefill_ms, 'score_forwards(reward, actor, reference, critic)': score_ms,
                     'rl_step(actor + critic fwd/bwd/clip/AdamW)': rl_ms},
        'rollout_tokens_per_s': prompts * new_tokens / (gen_ms / 1e3), 'decode_ms_per_position': decode_ms,
        'decode_weight_stream_frac_of_hbm_peak': stream_bytes / (decode_ms / 1e3) / 8e12,
        'memory': {'resident_after_build_GiB': resident / 2 ** 30, 'peak_allocated_GiB': torch.cuda.max_memory_allocated(dev) / 2 ** 30,
                   'peak_reserved_GiB': torch.cuda.max_memory_reserved(dev) / 2 ** 30, 'device_total_GiB': torch.cuda.get_device_properties(dev).total_memory / 2 ** 30,
                   'weights_bf16_all_four_models_GB': weights_gb,
                   'decode_copies_resident_during_update': getattr(tr.actor_model.module.stack, '_dw', None) is not None, 'actor_trainable_params': st_a.num_trainable(), 'critic_trainable_params': st_c.num_trainable()},
        'build_s': build_s, 'iterations': rows,
    }
    return out


if __name__ == '__main__':
    main()
