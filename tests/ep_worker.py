"""Worker for tests/test_ep_gpu.py: rank r of a 2-rank job holds experts [4r, 4r + 4) of the tiny Qwen3-MoE fixture and runs ONE
native DPO train_step on pair r with `expert_parallel: true` (all ranks share cuda:0, gloo collectives with host-staged
all-to-all -- the 1-GPU box has no second device for RCCL).  Rank 0 saves loss, gradients (dense: after the all-reduce;
experts: the owners' shards put back together) and the updated weights in HF layout.  Launched by torch.distributed.run."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import load_golden, state_dict_from_golden, tiny_qwen3moe_cfg  # noqa: E402


def run(z, dtype, rank):
    from align_anything_amd.trainers.dpo import DPOTrainer
    cfgs = {'train_cfgs': {'scale_coeff': float(z['scale_coeff']), 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant',
                           'weight_decay': 0.0, 'compute_dtype': dtype, 'expert_parallel': True},
            'model_cfgs': {'pad_token_id': int(z['pad_token_id'])}}
    wd = torch.bfloat16 if dtype == 'bf16' else torch.float32
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=tiny_qwen3moe_cfg(), policy_state=state_dict_from_golden(z, 'w.', wd),
                    reference_state=state_dict_from_golden(z, 'r.', wd), device='cuda:0')
    st = tr.policy.store
    assert tr.model.world == 2 and tr.policy.ep.size == 2 and 'exp' in st.gflat
    assert st.p['model.layers.0.mlp.experts.gate_up_proj'].shape[0] == 4
    rows = [rank, rank + 2]          # pair i = (chosen i, rejected i)
    T = torch.from_numpy
    mb = {'input_ids': T(z['input_ids'])[rows].cuda(), 'attention_mask': T(z['attention_mask'])[rows].cuda(),
          'meta_info': {'response_lens': [int(z['response_lens'][r]) for r in rows]}}
    info = tr.train_step(mb)
    tr.model.wait_optimizer()
    torch.cuda.synchronize()
    grads = {}
    for name in st.hf_names():
        g = st.grad_view(name)
        if g is None:
            continue
        g = g.float()
        if name in st.shard:
            g = tr.policy.ep.all_gather_rows(g)
        grads[name] = (g * 0.5).cpu()            # sum over the two ranks' local-mean losses -> gradient of the batch mean
    sd = {k: v.float().cpu() for k, v in tr.policy.state_dict().items()}       # collective: experts gathered
    gn = tr.model.grad_norm()
    return {'info': info, 'grads': grads, 'state': sd, 'grad_norm': gn}


def main():
    out = sys.argv[1]
    rank = int(os.environ['RANK'])
    torch.cuda.set_device(0)
    dist.init_process_group('gloo')
    z = load_golden('qwen3moe_tiny_dpo.npz')
    res = {dt: run(z, dt, rank) for dt in ('fp32', 'bf16')}
    sums = [None, None]
    dist.all_gather_object(sums, {dt: (r['grad_norm'], float(r['state']['model.norm.weight'].double().sum()),
                                       float(r['state']['model.layers.1.mlp.experts.down_proj'].double().sum())) for dt, r in res.items()})
    if rank == 0:
        assert sums[0] == sums[1], f'ranks disagree on clip norm / replicated weights after the step: {sums}'
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
