"""GPU: expert-parallel Qwen3-MoE step on hardware (align_anything_amd/expert_parallel.py; SURVEY.md section 8f rank 4 -- new design, the
reference has no counterpart, so the oracle is the SAME native step without expert parallelism, itself pinned to the reference's
fixture by tests/test_qwen3moe_gpu.py).  Two ranks, four of the eight experts each, one pair each, must reproduce ONE rank stepping on
the 2-pair batch with all experts: loss, every gradient (replicated ones after the all-reduce, expert shards untouched by it), the
clip norm and the updated weights."""
import os
import subprocess
import sys

import pytest
import torch

from tests.gpu_util import dev, dump
from tests.util import ROOT, load_golden, rel_err, state_dict_from_golden, tiny_qwen3moe_cfg

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _full_batch_step(z, dtype):
    from align_anything_amd.trainers.dpo import DPOTrainer
    cfgs = {'train_cfgs': {'scale_coeff': float(z['scale_coeff']), 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant',
                           'weight_decay': 0.0, 'compute_dtype': dtype},
            'model_cfgs': {'pad_token_id': int(z['pad_token_id'])}}
    wd = torch.bfloat16 if dtype == 'bf16' else torch.float32
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=tiny_qwen3moe_cfg(), policy_state=state_dict_from_golden(z, 'w.', wd),
                    reference_state=state_dict_from_golden(z, 'r.', wd), device='cuda:0')
    b = {'input_ids': T(z['input_ids']).to(dev()), 'attention_mask': T(z['attention_mask']).to(dev()),
         'meta_info': {'response_lens': [int(x) for x in z['response_lens']]}}
    info = tr.train_step(b)
    tr.model.wait_optimizer()
    torch.cuda.synchronize()
    st = tr.policy.store
    grads = {n: st.grad_view(n).float().cpu() for n in st.hf_names() if st.grad_view(n) is not None}
    return info, grads, {k: v.float().cpu() for k, v in tr.policy.state_dict().items()}, tr.model.grad_norm()


def test_two_rank_expert_parallel_step_equals_single_rank(tmp_path):
    out = str(tmp_path / 'ep2.pt')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29577', os.path.join(ROOT, 'tests', 'ep_worker.py'), out]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        dump('ep_worker_failure.log', r.stdout + '\n' + r.stderr)
    assert r.returncode == 0, r.stderr[-3000:]
    ep = torch.load(out)
    z = load_golden('qwen3moe_tiny_dpo.npz')
    rep = []
    for dtype, (g_tol, l_tol) in (('fp32', (2e-4, 2e-5)), ('bf16', (1.2e-1, 2e-2))):
        info, grads, state, gnorm = _full_batch_step(z, dtype)
        e = ep[dtype]
        assert abs(e['info']['train/loss'] - info['train/loss']) < l_tol, (dtype, e['info']['train/loss'], info['train/loss'])
        assert abs(e['grad_norm'] - gnorm) < (1e-4 if dtype == 'fp32' else 5e-2) * max(gnorm, 1.0), (dtype, e['grad_norm'], gnorm)
        worst, n, n_exp = 0.0, 0, 0
        assert set(e['grads']) == set(grads)
        for name, want in grads.items():
            got = e['grads'][name].reshape(want.shape)
            if float(want.norm()) < 1e-6:
                assert float(got.norm()) < 1e-4, name
                continue
            err = rel_err(got, want)
            worst = max(worst, err); n += 1; n_exp += 'experts' in name
            assert err < g_tol, (dtype, name, err)
        assert n >= 25 and n_exp >= 4
        # one AdamW step moves every weight by ~lr; a sign flip of a near-zero gradient component costs 2 lr (as in test_dp_gpu)
        for name, want in state.items():
            d = (e['state'][name].reshape(want.shape) - want).abs()
            assert d.max().item() <= 2.1e-3 + (0 if dtype == 'fp32' else 8e-3), (dtype, name, d.max().item())
            if dtype == 'fp32':
                assert (d < 1e-4).float().mean().item() > 0.97, (name, (d < 1e-4).float().mean().item())
        rep.append(f'{dtype}: loss ep {e["info"]["train/loss"]:.6f} single {info["train/loss"]:.6f}; clip norm ep {e["grad_norm"]:.6f} single {gnorm:.6f}; '
                   f'worst gradient rel_err {worst:.2e} over {n} tensors ({n_exp} expert tensors)')
    # the sync-free capacity-padded exchange (expert_parallel.py; trainers' default) is the SAME computation: bit-identical to the exact exchange
    for dtype in ('fp32', 'bf16'):
        e, pd = ep[dtype], ep['padded'][dtype]
        assert pd['info']['train/loss'] == e['info']['train/loss'] and pd['grad_norm'] == e['grad_norm'], (dtype, pd['info'], e['info'])
        n_same = 0
        for name, want in e['grads'].items():
            got = pd['grads'][name]
            same = torch.equal(got, want)
            n_same += same
            # the MoE block (what the exchange touches) and every GEMM-produced gradient: bit for bit.  Gradients summed with fp32 atomics
            # (embedding rows, norm weights) are not reproducible run to run even on ONE code path: last-bit tolerance there
            assert same or ('experts' not in name and 'mlp.gate' not in name and rel_err(got, want) < 1e-6), (dtype, name, float((got - want).abs().max()))
        for name, want in e['state'].items():
            assert torch.equal(pd['state'][name], want) or float((pd['state'][name] - want).abs().max()) <= 1e-6, (dtype, name)
        rep.append(f'{dtype}: capacity-padded exchange (factor 2.0, no host read per block) vs exact exchange: loss and clip norm identical, '
                   f'{n_same} of {len(e["grads"])} gradients bit-identical (all expert / router tensors; the rest differ in the last bit of fp32 atomic sums)')
    # blocks of exactly the balanced share: the fixture's router is not perfectly balanced, so the step must be reported invalid
    raised = [t['overflow_raised'] for t in ep['tight']]
    assert any(raised) or all(t['loss'] == ep['bf16']['info']['train/loss'] for t in ep['tight']), ep['tight']
    rep.append(f'capacity factor 1.0 (blocks of exactly the balanced share): overflow reported on ranks {[r for r, x in enumerate(raised) if x]}')
    # rollout under expert parallelism (generation.py: lockstep passes, token exchange per position) == the all-experts rollout
    for r, roll in enumerate(ep['rollout']):
        for key in ('no_eos', 'eos'):
            got_shape, want_shape, same = roll[key]
            assert same and got_shape == want_shape, (r, key, roll)
        assert roll['grpo_rollout_rows'] == 4
        rep.append(f'rank {r}: expert-parallel greedy rollout == all-experts rollout, shapes {roll["no_eos"][0]} / with EOS {roll["eos"][0]}')
    dump('parity_expert_parallel.txt', '\n'.join(rep) + '\n')
