"""Diagnostic (GPU box): native fp32 DPO steps vs a torch-CPU AdamW loop on the same HF OPT-125m weights.
Prints, per step, loss and per-tensor gradient / weight-update agreement."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from transformers import get_scheduler
from oracle.synthetic import opt125m_config1
from align_anything_amd import configs
from align_anything_amd.trainers.dpo import DPOTrainer

NSTEP = int(sys.argv[1]) if len(sys.argv) > 1 else 4
oc, policy, refm, batches = opt125m_config1(num_pairs=NSTEP)
steps = 64
cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-6, 'lr_warmup_ratio': 0.03, 'lr_scheduler_type': 'cosine', 'weight_decay': 0.05,
                       'adam_betas': [0.9, 0.95], 'total_training_steps': steps, 'compute_dtype': 'fp32'},
        'model_cfgs': {'pad_token_id': oc.pad_token_id}}
tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=configs.from_hf_config(oc), policy_state=policy.state_dict(),
                reference_state=refm.state_dict(), device='cuda:0')
tr.model.async_optimizer = False
nd = ['bias', 'layer_norm.weight', 'layernorm.weight', 'norm.weight', 'ln_f.weight']
named = dict(policy.named_parameters())
opt = torch.optim.AdamW([{'params': [p for n, p in named.items() if not any(x in n for x in nd)], 'weight_decay': 0.05},
                         {'params': [p for n, p in named.items() if any(x in n for x in nd)], 'weight_decay': 0.0}], lr=1e-6, betas=(0.9, 0.95), eps=1e-8)
sched = get_scheduler('cosine', opt, num_warmup_steps=int(0.03 * steps), num_training_steps=steps)
torch.set_num_threads(8)


def seq_logp(model, ids, R):
    logits = model(input_ids=ids, attention_mask=torch.ones_like(ids)).logits
    lp = torch.log_softmax(logits[:, -R:][:, :-1].float(), -1).gather(-1, ids[:, -R:][:, 1:].unsqueeze(-1)).squeeze(-1)
    return lp.sum(-1)


def seq_logp64(model, ids, R):
    logits = model(input_ids=ids, attention_mask=torch.ones_like(ids)).logits
    lp = torch.log_softmax(logits[:, -R:][:, :-1], -1).gather(-1, ids[:, -R:][:, 1:].unsqueeze(-1)).squeeze(-1)
    return lp.sum(-1)


st = tr.policy.store
for i, b in enumerate(batches):
    ids = b['input_ids']; R = 128
    lp = seq_logp(policy, ids, R)
    with torch.no_grad():
        rlp = seq_logp(refm, ids, R)
    loss = -F.logsigmoid(0.1 * ((lp[0] - rlp[0]) - (lp[1] - rlp[1])))
    loss.backward()
    w_before = {n: st.view(n).detach().float().cpu().clone() for n in ('model.decoder.layers.3.fc1.weight', 'model.decoder.embed_tokens.weight', 'model.decoder.layers.0.self_attn.q_proj.bias', 'model.decoder.final_layer_norm.weight')}
    cpu_before = {n: named[n].detach().clone() for n in w_before}
    info = tr.train_step({'input_ids': ids.cuda(), 'attention_mask': torch.ones_like(ids).cuda(), 'meta_info': b['meta_info']})
    torch.cuda.synchronize()
    print(f'step {i}: native loss {info["train/loss"]:.6f} cpu loss {loss.item():.6f}  native gnorm {tr.model.grad_norm():.5f}', end='')
    worst = 0
    errs = []
    for n, p in named.items():
        g = st.grad_view(n)
        e = float((g.float().cpu().reshape(p.grad.shape) - p.grad).norm() / (p.grad.norm() + 1e-30))
        errs.append((e, n, float(p.grad.norm())))
        worst = max(worst, e)
    print()
    if i == 0:
        import copy
        p64 = copy.deepcopy(policy).double(); r64 = copy.deepcopy(refm).double()
        p64.zero_grad()
        lp64 = seq_logp64(p64, ids, R)
        with torch.no_grad():
            rlp64 = seq_logp64(r64, ids, R)
        (-F.logsigmoid(0.1 * ((lp64[0] - rlp64[0]) - (lp64[1] - rlp64[1])))).backward()
        g64 = {n: q.grad for n, q in p64.named_parameters()}
        for L in (11, 10, 8, 6, 4, 2, 0):
            for suf in ('self_attn.v_proj.weight', 'self_attn.q_proj.weight', 'self_attn.out_proj.weight', 'fc2.weight', 'fc1.weight', 'self_attn_layer_norm.weight'):
                n = f'model.decoder.layers.{L}.{suf}'
                t = g64[n]
                en = float((st.grad_view(n).double().cpu().reshape(t.shape) - t).norm() / t.norm())
                ec = float((named[n].grad.double() - t).norm() / t.norm())
                print(f'      L{L:2d} {suf:28s} native-vs-fp64 {en:.2e}   torchcpu-fp32-vs-fp64 {ec:.2e}')
    gn = torch.nn.utils.clip_grad_norm_(policy.parameters(), 1.0)
    print(f' cpu gnorm {float(gn):.5f} worst grad rel {worst:.2e}')
    opt.step(); sched.step(); opt.zero_grad(set_to_none=True)
    for n in w_before:
        dn = st.view(n).detach().float().cpu() - w_before[n]
        dc = named[n].detach() - cpu_before[n]
        print(f'    {n}: |dW| native {float(dn.abs().mean()):.3e} cpu {float(dc.abs().mean()):.3e}  rel diff of update {float((dn.reshape(dc.shape) - dc).norm() / (dc.norm() + 1e-30)):.3e}')
