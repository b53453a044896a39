"""Lab (round 6): which python lines launch the torch copy / fill kernels inside one headline DPO step.
    python tools/lab/glue_prof.py [layers]      -> gpurun_out/r06_glue_prof.txt"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from align_anything_amd import configs  # noqa: E402
from align_anything_amd.trainers.dpo import DPOTrainer  # noqa: E402

MOE = len(sys.argv) > 1 and sys.argv[1] == 'moe'          # `glue_prof.py moe`: the Qwen3-MoE step of tools/bench_qwen3moe.py (12 layers, 2 pairs) -> r06_glue_prof_moe.txt
layers = 12 if MOE else (int(sys.argv[1]) if len(sys.argv) > 1 else 8)
dev = torch.device('cuda:0')
B, T, R = (2 if MOE else 4), 2048, 512
if MOE:
    cfg = configs.qwen3moe_cfg(2048, 768, layers, 32, 4, 151936, 128, 8, head_dim=128)
    cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-6, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant', 'weight_decay': 0.0}, 'model_cfgs': {'pad_token_id': 0}}
else:
    cfg = configs.llava_1_5_7b(num_layers=layers)
    cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-6, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant', 'freeze_mm_proj': False, 'freeze_language_model': False, 'freeze_vision_tower': True}, 'model_cfgs': {'pad_token_id': cfg['pad_token_id']}}
tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, device=dev)
bench.random_init_(tr.policy, seed=42)
tr.reference.load_state_dict(tr.policy.state_dict())
for g_ in tr.policy.store.master:
    tr.policy.store.master[g_].copy_(tr.policy.store.flat[g_])


def _moe_batch(seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, 151000, (2 * B, T), generator=g)
    ids[B:, :T - R] = ids[:B, :T - R]
    return {'input_ids': ids.to(dev), 'attention_mask': torch.ones(2 * B, T, dtype=torch.long, device=dev), 'meta_info': {'response_lens': [R] * (2 * B)}}


batches = [_moe_batch(100 + i) if MOE else bench.make_batch(cfg, B, T, R, dev, seed=100 + i) for i in range(4)]
for i in range(2):
    tr.train_step(batches[i])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(batches[2])
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name.startswith('aten::') and ev.device_time_total > 0 and not any(c.name.startswith('aten::') and c.device_time_total > 0 for c in ev.cpu_children):
        site = next((s for s in ev.stack if 'align_anything_amd' in s or 'bench.py' in s), ev.stack[0] if ev.stack else '?')
        k = (ev.name, site.split('align_anything_amd/')[-1][:90])
        agg[k][0] += 1
        agg[k][1] += ev.device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
out = [f'{layers} layers, B = {B}: torch ops inside ONE train_step, by python site (count, device us)']
tot = 0.0
for (name, site), (n, us) in rows[:60]:
    out.append(f'{us:9.1f} us  {n:4d} x  {name:18s} {site}')
for (_, _), (n, us) in rows:
    tot += us
out.append(f'total device time of these ops: {tot / 1e3:.2f} ms')
kern = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA and ('rocclr' in ev.name or 'at::native' in ev.name):
        kern[ev.name[:80]][0] += 1
        kern[ev.name[:80]][1] += ev.device_time_total
out.append('torch / runtime kernels on the device in that step:')
for k, (n, us) in sorted(kern.items(), key=lambda kv: -kv[1][1])[:15]:
    out.append(f'{us:9.1f} us  {n:4d} x  {k}')
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
open(os.path.join(ROOT, 'gpurun_out', 'r06_glue_prof_moe.txt' if MOE else 'r06_glue_prof.txt'), 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
