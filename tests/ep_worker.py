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


def run(z, dtype, rank, capacity_factor=0.0):
    """capacity_factor 0: the exact exchange (one host read per block); 2.0: the sync-free capacity-padded one (on 2 ranks a block holds every
    pair, so it cannot overflow, and about half of every block is the zero tail); 1.0: blocks of exactly the balanced share -- any imbalance of
    the router overflows, which must surface as an error at the blocking poll."""
    from align_anything_amd.trainers.dpo import DPOTrainer
    cfgs = {'train_cfgs': {'scale_coeff': float(z['scale_coeff']), 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant',
                           'weight_decay': 0.0, 'compute_dtype': dtype, 'expert_parallel': True, 'expert_parallel_capacity_factor': capacity_factor,
                           'expert_parallel_dense_below': 0},
            'model_cfgs': {'pad_token_id': int(z['pad_token_id'])}}
    wd = torch.bfloat16 if dtype == 'bf16' else torch.float32
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=tiny_qwen3moe_cfg(), policy_state=state_dict_from_golden(z, 'w.', wd),
                    reference_state=state_dict_from_golden(z, 'r.', wd), device='cuda:0')
    st = tr.policy.store
    assert tr.model.world == 2 and tr.policy.ep.size == 2 and 'exp' in st.gflat and tr.policy.ep.padded == (capacity_factor > 0)
    assert st.p['model.layers.0.mlp.experts.gate_up_proj'].shape[0] == 4
    rows = [rank, rank + 2]          # pair i = (chosen i, rejected i)
    T = torch.from_numpy
    mb = {'input_ids': T(z['input_ids'])[rows].cuda(), 'attention_mask': T(z['attention_mask'])[rows].cuda(),
          'meta_info': {'response_lens': [int(z['response_lens'][r]) for r in rows]}}
    info = tr.train_step(mb)
    tr.model.wait_optimizer()
    torch.cuda.synchronize()
    if capacity_factor == 1.0:
        try:
            tr.model.grad_norm()
        except RuntimeError as e:
            return {'overflow_raised': 'capacity' in str(e)}
        return {'overflow_raised': False, 'loss': info['train/loss']}
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


def rollout(z, rank):
    """Greedy rollout with expert-parallel weights (token exchange per decode position, ranks in lockstep) == the same rollout on a
    replica holding all experts.  The ranks get prompts of DIFFERENT lengths under one max_length (so their own step budgets differ)
    and, in the second run, an EOS that ends rank-local rows early: the pass count must still agree on every rank."""
    from align_anything_amd.expert_parallel import ExpertParallel
    from align_anything_amd.generation import generate
    from align_anything_amd.modeling import build_model
    cfg = tiny_qwen3moe_cfg()
    sd = state_dict_from_golden(z, 'w.', torch.bfloat16)
    ep_model = build_model(cfg, 'cuda:0', trainable=False, ep=ExpertParallel(dist.new_group()))
    ep_model.load_state_dict(sd)
    full = build_model(cfg, 'cuda:0', trainable=False)
    full.load_state_dict(sd)
    L = 10 if rank == 0 else 14
    ids = torch.from_numpy(z['input_ids'])[[rank, rank + 2], :L].clone().cuda()
    am = torch.ones_like(ids)
    am[1, :2] = 0                                            # a left-padded row
    ids[1, :2] = int(z['pad_token_id'])
    kw = dict(max_length=24, do_sample=False, pad_token_id=int(z['pad_token_id']))
    rep = {}
    want = generate(full, ids, am, eos_token_id=None, **kw)
    got = generate(ep_model, ids, am, eos_token_id=None, **kw)
    rep['no_eos'] = (tuple(got.shape), tuple(want.shape), bool(torch.equal(got, want)))
    eos = int(want[0, L + 2])                                # ends row 0 at its third new token (other rows: whenever they emit it)
    want = generate(full, ids, am, eos_token_id=eos, sync_every=2, **kw)
    got = generate(ep_model, ids, am, eos_token_id=eos, sync_every=2, **kw)
    rep['eos'] = (tuple(got.shape), tuple(want.shape), bool(torch.equal(got, want)))
    # the trainers' flag: a GRPO trainer with train_cfgs.expert_parallel builds sharded actor / reference / reward models and rolls out
    from align_anything_amd.trainers.grpo import GRPOTrainer
    cfgs = {'train_cfgs': {'expert_parallel': True, 'num_generations': 2, 'actor_lr_scheduler_type': 'constant'},
            'model_cfgs': {'pad_token_id': int(z['pad_token_id']), 'eos_token_id': eos, 'model_max_length': 20}}
    tr = GRPOTrainer(cfgs, {}, model_cfg=cfg, actor_state=sd, reward_fn=lambda completions: torch.zeros(completions.shape[0]), device='cuda:0')
    assert tr.actor_model.module.ep is not None and tr.actor_reference_model.module.ep is not None
    seqs = tr.generate_completions({'input_ids': ids, 'attention_mask': am}, generator=torch.Generator(device='cuda').manual_seed(5 + rank))
    rep['grpo_rollout_rows'] = int(seqs.shape[0])
    return rep


def main():
    out = sys.argv[1]
    rank = int(os.environ['RANK'])
    torch.cuda.set_device(0)
    import datetime
    # a rank that dies must fail its peer within minutes, not after gloo's default half hour (the GPU suite runs under the driver's clock)
    dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=240))
    z = load_golden('qwen3moe_tiny_dpo.npz')
    res = {dt: run(z, dt, rank) for dt in ('fp32', 'bf16')}
    padded = {dt: run(z, dt, rank, capacity_factor=2.0) for dt in ('fp32', 'bf16')}
    tight = [None, None]
    dist.all_gather_object(tight, run(z, 'bf16', rank, capacity_factor=1.0))
    rolls = [None, None]
    dist.all_gather_object(rolls, rollout(z, rank))
    sums = [None, None]
    dist.all_gather_object(sums, {dt: (r['grad_norm'], float(r['state']['model.norm.weight'].double().sum()),
                                       float(r['state']['model.layers.1.mlp.experts.down_proj'].double().sum())) for dt, r in res.items()})
    if rank == 0:
        assert sums[0] == sums[1], f'ranks disagree on clip norm / replicated weights after the step: {sums}'
        res['rollout'] = rolls
        res['padded'] = padded
        res['tight'] = tight
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
