"""GPU: the fp32 PARITY MODE (include/aa_hip_f32.h; dtype=torch.float32 on the native models / trainers).

Kernel level: the fp32 twins against fp64 torch references at fp32 tolerances.
Model level: tiny LLaVA / OPT DPO against the fixtures the REFERENCE's own DPOTrainer produced in fp32 -- the same
fixtures the bf16 path meets at ~1e-2 are met here at ~1e-5.
Headline: BASELINE.json's "loss curves matching reference to 1e-4" on configs[0] (OPT-125m, 64 pairs, seq 256):
64 native train_steps against the curve of the reference's unmodified DPOTrainer.train_step (tests/golden/
opt125m_curve.npz, oracle/gen_golden.py::gen_opt125m_curve)."""
import numpy as np
import pytest
import torch

from tests.gpu_util import dev, dump
from tests.util import load_golden, rel_err, state_dict_from_golden, tiny_llava_cfg, tiny_opt_cfg

pytestmark = pytest.mark.gpu
T_ = torch.from_numpy
F32 = torch.float32


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev())


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize('layout', ['nt', 'nn', 'tn'])
@pytest.mark.parametrize('M,N,K', [(200, 136, 192), (64, 64, 64), (513, 260, 128), (1, 4, 16)])
def test_gemm_f32_layouts(layout, M, N, K):
    from align_anything_amd import ops
    if layout == 'tn':
        M = (M + 3) // 4 * 4          # A stored [K][M]: its leading dimension M must be a multiple of 4 floats
    a = rnd(M, K, seed=1); b = rnd(N, K, seed=2)
    ref = (a.double() @ b.double().t())
    if layout == 'nt':
        out = ops.gemm(a, b)
    elif layout == 'nn':
        out = ops.gemm(a, b.t().contiguous(), b_n=True)
    else:
        out = ops.gemm(a.t().contiguous(), b.t().contiguous(), a_t=True, b_n=True)
    assert out.dtype == F32 and rel_err(out, ref) < 2e-6, rel_err(out, ref)


def test_gemm_f32_epilogue_bias_act_residual_accumulate_and_views():
    from align_anything_amd import ops
    M, N, K = 130, 72, 64
    a, w, bias, res = rnd(M, K, seed=3), rnd(N, K, seed=4), rnd(N, seed=5), rnd(M, N, seed=6)
    for act, fn in ((ops.ACT_NONE, lambda x: x), (ops.ACT_RELU, torch.relu), (ops.ACT_GELU, torch.nn.functional.gelu),
                    (ops.ACT_SILU, torch.nn.functional.silu), (ops.ACT_QUICK_GELU, lambda x: x * torch.sigmoid(1.702 * x))):
        ref = fn(a.double() @ w.double().t() + bias.double()) + res.double()
        out = ops.gemm(a, w, bias=bias, residual=res, act=act)
        assert rel_err(out, ref) < 3e-6, (act, rel_err(out, ref))
    acc = rnd(M, N, seed=7)
    ref = acc.double() + a.double() @ w.double().t()
    ops.gemm(a, w, out=acc, accumulate=True)
    assert rel_err(acc, ref) < 3e-6
    # column-slice views (fused qkv buffers) as operands and output
    big = rnd(M, 3 * K, seed=8)
    outbig = torch.zeros(M, 2 * N, device=dev())
    ops.gemm(big[:, K:2 * K], w, out=outbig[:, N:])
    assert rel_err(outbig[:, N:], big[:, K:2 * K].double() @ w.double().t()) < 3e-6 and float(outbig[:, :N].abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        ops.gemm(a, w.to(torch.bfloat16))


# ------------------------------------------------------------------------------------------------ attention
def ref_attention64(q, k, v, do, N, T, H, Hkv, hd, causal, scale, start):
    qf = q.double().view(N, T, H, hd).transpose(1, 2).detach().requires_grad_(True)
    kf = k.double().view(N, T, Hkv, hd).transpose(1, 2).detach().requires_grad_(True)
    vf = v.double().view(N, T, Hkv, hd).transpose(1, 2).detach().requires_grad_(True)
    rep = H // Hkv
    s = (qf @ kf.repeat_interleave(rep, 1).transpose(-1, -2)) * scale
    idx = torch.arange(T, device=q.device)
    mask = torch.zeros(N, 1, T, T, dtype=torch.bool, device=q.device)
    if causal:
        mask = mask | (idx[None, :] > idx[:, None])[None, None]
    valid = torch.ones(N, T, dtype=torch.bool, device=q.device)
    if start is not None:
        valid = idx[None, :] >= start[:, None].long()
        mask = mask | ~valid[:, None, None, :]
    s = s.masked_fill(mask, float('-inf'))
    row_ok = ~mask.all(-1, keepdim=True)
    p = torch.softmax(s.masked_fill(~row_ok, 0.0), -1) * row_ok
    o = p @ vf.repeat_interleave(rep, 1)
    lse = torch.logsumexp(s.masked_fill(~row_ok, 0.0), -1)
    dof = do.double().view(N, T, H, hd).transpose(1, 2) * valid[:, None, :, None]
    (o * dof).sum().backward()
    back = lambda t, h: t.transpose(1, 2).reshape(N * T, h * hd)
    return back(o.detach(), H), back(qf.grad, H), back(kf.grad, Hkv), back(vf.grad, Hkv), valid.reshape(N * T), lse.detach(), row_ok.squeeze(-1).expand(N, H, T)


@pytest.mark.parametrize('case', [(2, 256, 2, 2, 128, True, [0, 37]), (1, 200, 2, 2, 128, True, [70]), (2, 577, 2, 2, 64, False, None),
                                  (2, 128, 4, 2, 64, True, [0, 5]), (3, 48, 2, 1, 64, True, [0, 6, 2]), (1, 33, 1, 1, 128, False, [4])])
def test_attention_f32_forward_backward(case):
    from align_anything_amd import ops
    N, T, H, Hkv, hd, causal, starts = case
    scale = hd ** -0.5
    qkv = rnd(N * T, (H + 2 * Hkv) * hd, seed=11)
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:(H + Hkv) * hd], qkv[:, (H + Hkv) * hd:]
    do = rnd(N * T, H * hd, seed=12)
    start = torch.tensor(starts, dtype=torch.int32, device=dev()) if starts is not None else None
    ro, rdq, rdk, rdv, valid, rlse, row_ok = ref_attention64(q, k, v, do, N, T, H, Hkv, hd, causal, scale, start)
    o, lse = ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, causal, scale, start)
    assert o.dtype == F32
    dqkv = torch.zeros_like(qkv)
    dq, dk, dv = dqkv[:, :H * hd], dqkv[:, H * hd:(H + Hkv) * hd], dqkv[:, (H + Hkv) * hd:]
    do = do * valid[:, None].to(do.dtype)
    ops.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, N, T, H, Hkv, hd, causal, scale, start)
    torch.cuda.synchronize()
    vm = valid[:, None]
    dead = ~row_ok.transpose(1, 2).reshape(N * T, H)                    # (row, head) with no attendable key
    assert float((o.view(N * T, H, hd) * dead[:, :, None]).abs().max()) == 0.0, 'fully masked query rows must be exactly 0'
    assert rel_err(o * vm, ro * vm) < 3e-6
    assert torch.allclose(lse[row_ok].double(), rlse[row_ok], rtol=1e-6, atol=2e-6)
    assert torch.isinf(lse[~row_ok]).all()
    assert rel_err(dq * vm, rdq * vm) < 1e-5 and rel_err(dk, rdk) < 1e-5 and rel_err(dv, rdv) < 1e-5, \
        (rel_err(dq * vm, rdq * vm), rel_err(dk, rdk), rel_err(dv, rdv))


# ------------------------------------------------------------------------------------------------ block kernels
def test_block_kernel_twins_match_torch_fp64():
    """A representative sweep of the second instantiation of elementwise.hip (same source, elem_t = float)."""
    from align_anything_amd import ops
    rows, h = 70, 256
    x, w, b, dy = rnd(rows, h, seed=1), rnd(h, seed=2) * 0.3 + 1, rnd(h, seed=3) * 0.1, rnd(rows, h, seed=4)
    # RMSNorm
    xd = x.double().requires_grad_(True); wd = w.double().requires_grad_(True)
    y_ref = xd * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-5) * wd
    y_ref.backward(dy.double())
    y, rstd = ops.rmsnorm_fwd(x, w, 1e-5)
    dw = torch.zeros(h, device=dev())
    dx = ops.rmsnorm_bwd(dy, x, w, rstd, dw)
    assert rel_err(y, y_ref) < 1e-6 and rel_err(dx, xd.grad) < 3e-6 and rel_err(dw, wd.grad) < 3e-6
    # LayerNorm
    xd = x.double().requires_grad_(True); wd = w.double().requires_grad_(True); bd = b.double().requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm(xd, (h,), wd, bd, 1e-5)
    y_ref.backward(dy.double())
    y, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-5)
    dw, db = torch.zeros(h, device=dev()), torch.zeros(h, device=dev())
    dx = ops.layernorm_bwd(dy, x, w, mean, rstd, dw, db)
    assert rel_err(y, y_ref) < 1e-6 and rel_err(dx, xd.grad) < 5e-6 and rel_err(dw, wd.grad) < 3e-6 and rel_err(db, bd.grad) < 3e-6
    # SwiGLU
    gu, da = rnd(rows, 2 * h, seed=5), rnd(rows, h, seed=6)
    gd = gu.double().requires_grad_(True)
    a_ref = torch.nn.functional.silu(gd[:, :h]) * gd[:, h:]
    a_ref.backward(da.double())
    assert rel_err(ops.swiglu_fwd(gu), a_ref) < 1e-6 and rel_err(ops.swiglu_bwd(gu, da), gd.grad) < 3e-6
    # RoPE forward + inverse is the identity; forward equals the half-split rotation
    hd, nh = 64, 2
    buf = rnd(rows, nh * hd, seed=7); orig = buf.clone()
    pos = torch.arange(rows, dtype=torch.int32, device=dev())
    from align_anything_amd.modeling import rope_tables
    cos, sin = rope_tables(128, hd, 10000.0, dev(), F32)
    ops.rope_(buf, 0, nh, hd, pos, cos, sin)
    xv = orig.double().view(rows, nh, hd)
    c = torch.cat([cos[:rows], cos[:rows]], -1).double()[:, None]; s_ = torch.cat([sin[:rows], sin[:rows]], -1).double()[:, None]
    rot = torch.cat([-xv[..., hd // 2:], xv[..., :hd // 2]], -1)
    assert rel_err(buf.view(rows, nh, hd), xv * c + rot * s_) < 1e-6
    ops.rope_(buf, 0, nh, hd, pos, cos, sin, inverse=True)
    assert rel_err(buf, orig) < 1e-6
    # embedding gather/scatter, transpose, column sum, add, activation, score head
    E = rnd(50, h, seed=8); ids = torch.randint(0, 50, (rows,), device=dev())
    assert torch.equal(ops.embed_fwd(ids, E), E[ids])
    dE = torch.zeros(50, h, device=dev())
    ops.embed_bwd(ids, dy, 50, dE=dE)
    assert rel_err(dE, torch.zeros(50, h, device=dev(), dtype=torch.float64).index_add_(0, ids, dy.double())) < 1e-6
    assert torch.equal(ops.transpose(x), x.t().contiguous())
    cs = torch.zeros(h, device=dev()); ops.colsum_(x, cs)
    assert rel_err(cs, x.double().sum(0)) < 1e-6
    assert torch.equal(ops.add(x, dy), x + dy)
    assert rel_err(ops.act_bwd(x, dy, ops.ACT_GELU), torch.autograd.grad(torch.nn.functional.gelu(xq := x.double().requires_grad_(True)), xq, dy.double())[0]) < 3e-6
    sc = ops.rowdot_fwd(x, w)
    assert rel_err(sc, x.double() @ w.double()) < 1e-6


# ------------------------------------------------------------------------------------------------ model level
def _trainer(z, cfg, **train_cfgs):
    from align_anything_amd.trainers.dpo import DPOTrainer
    cfgs = {'train_cfgs': dict({'scale_coeff': float(z['scale_coeff']), 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant',
                                'weight_decay': 0.05, 'adam_betas': [0.9, 0.95], 'compute_dtype': 'fp32'}, **train_cfgs),
            'model_cfgs': {'pad_token_id': int(z['pad_token_id'])}}
    return DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, policy_state=state_dict_from_golden(z, 'w.'),
                      reference_state=state_dict_from_golden(z, 'r.'), device='cuda:0',
                      share_vision_tower=False)   # the fixture's reference model has its own (perturbed) tower


@pytest.mark.parametrize('name,cfg_fn,pixels', [('llava_tiny_dpo.npz', tiny_llava_cfg, True), ('opt_tiny_dpo.npz', tiny_opt_cfg, False)])
def test_fp32_mode_matches_reference_fixture_at_fp32_tolerance(name, cfg_fn, pixels):
    z = load_golden(name)
    tr = _trainer(z, cfg_fn())
    assert tr.policy.dtype == F32 and tr.policy.store.master['mat'] is tr.policy.store.flat['mat']
    b = {'input_ids': T_(z['input_ids']).to(dev()), 'attention_mask': T_(z['attention_mask']).to(dev()),
         'meta_info': {'response_lens': [int(x) for x in z['response_lens']]}}
    if pixels:
        b['pixel_values'] = T_(z['pixel_values']).to(dev())
    logits = tr.policy.logits(b['input_ids'], b['attention_mask'], b.get('pixel_values')).cpu()
    valid = T_(z['attention_mask']).bool()
    rep = [f'logits rel_err {rel_err(logits[valid], T_(z["policy_logits"])[valid]):.2e}']
    assert rel_err(logits[valid], T_(z['policy_logits'])[valid]) < 2e-5
    lp = tr.compute_log_probs(tr.model, b).cpu()
    assert torch.equal(lp == 0, T_(z['seq_log_probs']) == 0)
    assert (lp - T_(z['seq_log_probs'])).abs().max() < 5e-5
    ld = tr.loss(b)
    for k in ('loss', 'reward', 'better_sample_reward', 'worse_sample_reward', 'reward_accuracy', 'reward_margin'):
        got, want = ld[k].detach().float().cpu().reshape(-1), T_(z['loss_' + k]).float().reshape(-1)
        rep.append(f'{k}: native {got.tolist()} reference {want.tolist()}')
        assert (got - want).abs().max() < 2e-5, (k, got, want)
    tr.model.backward(ld['loss'])
    torch.cuda.synchronize()
    worst = 0.0
    for k in z.files:
        if not k.startswith('g.'):
            continue
        if k[2:].startswith('model.vision_tower'):
            continue  # frozen in the native trainer (reference default); the fixture has grads because HF ran unfrozen
        g = tr.policy.store.grad_view(k[2:])
        assert g is not None, k
        want = T_(z[k])
        got = g.float().cpu().reshape(want.shape)
        if float(want.norm()) < 1e-6:
            assert float(got.norm()) < 1e-5, k
            continue
        e = rel_err(got, want)
        worst = max(worst, e)
        assert e < 2e-4, (k, e)
    rep.append(f'worst gradient rel_err {worst:.2e}')
    dump(f'parity_fp32_{name.split("_")[0]}.txt', '\n'.join(rep) + '\n')


def test_opt125m_64_step_loss_curve_vs_reference():
    """BASELINE.json: 'loss curves matching reference to 1e-4' on configs[0]."""
    from oracle.synthetic import opt125m_config1            # HF init + torch CPU RNG only (no /root/reference access)
    from align_anything_amd import configs
    from align_anything_amd.trainers.dpo import DPOTrainer
    z = load_golden('opt125m_curve.npz')
    keys = [str(k) for k in z['keys']]
    gold = z['metrics']
    oc, policy, refm, batches = opt125m_config1()
    # the identical weights must have been regenerated (HF init from torch.manual_seed(42) on CPU)
    sd = policy.state_dict()
    for n, c in zip(z['checksum_names'], z['checksum']):
        assert abs(float(sd[str(n)].double().sum()) - float(c)) <= 1e-9 * max(1.0, abs(float(c))), f'weight init differs from the fixture: {n}'
    assert np.array_equal(batches[0]['input_ids'].numpy(), z['first_ids']) and np.array_equal(batches[-1]['input_ids'].numpy(), z['last_ids'])
    steps = len(batches)
    cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-6, 'lr_warmup_ratio': 0.03, 'lr_scheduler_type': 'cosine', 'weight_decay': 0.05,
                           'adam_betas': [0.9, 0.95], 'total_training_steps': steps, 'compute_dtype': 'fp32'},
            'model_cfgs': {'pad_token_id': oc.pad_token_id}}
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=configs.from_hf_config(oc), policy_state=sd,
                    reference_state=refm.state_dict(), device='cuda:0')
    rows = []
    for b in batches:
        info = tr.train_step({'input_ids': b['input_ids'].to(dev()), 'attention_mask': b['attention_mask'].to(dev()), 'meta_info': b['meta_info']})
        rows.append([info[k] for k in keys])
    got = np.array(rows, dtype=np.float64)
    li = keys.index('train/loss')
    err = np.abs(got[:, li] - gold[:, li])
    lines = [f'step {i:2d} native {got[i, li]:.6f} reference {gold[i, li]:.6f} |diff| {err[i]:.2e}' for i in range(steps)]
    # The reference cannot reproduce ITSELF to 1e-4 over 64 Adam steps: re-running the unmodified reference trainer with 3
    # CPU threads instead of 8 (same code, same seeds; only the fp32 summation order of the CPU GEMMs changes) moves its
    # curve by up to `self_dev` (stored beside the curve).  Cause: ReLU kinks -- a single pre-activation that rounds to the
    # other side of 0 changes a layer's gradient by ~1e-3 relative (torch fp32 vs fp64 gradients differ by 1e-3 from layer
    # 8 down) -- and Adam normalises every gradient element to a step of ~lr, so the two weight trajectories random-walk
    # apart.  The 1e-4 target is therefore asserted where it is well posed (the early steps, where the weights are still
    # common) and the whole curve is held to the reference's own reproducibility.
    self_dev = np.abs(z['metrics_3threads'][:, li] - gold[:, li])
    lines.append(f'max |loss diff| over {steps} steps = {err.max():.3e}; first 8 steps {err[:8].max():.3e} (target 1e-4)')
    lines.append(f'reference vs itself (3 vs 8 CPU threads): max |loss diff| = {self_dev.max():.3e}; first 8 steps {self_dev[:8].max():.3e}')
    for k in ('train/reward_margin', 'train/reward_accuracy', 'train/lr'):
        j = keys.index(k)
        lines.append(f'max |{k} diff| = {np.abs(got[:, j] - gold[:, j]).max():.3e}')
    dump('parity_fp32_opt125m_loss_curve.txt', '\n'.join(lines) + '\n')
    assert err[:8].max() < 1e-4, err[:8]
    assert err.max() < 2.5 * self_dev.max() and err.max() < 2e-3, (err.max(), self_dev.max())
    assert np.abs(got[:, keys.index('train/reward_margin')] - gold[:, keys.index('train/reward_margin')]).max() < 2e-2
    assert np.array_equal(got[:, keys.index('train/reward_accuracy')], gold[:, keys.index('train/reward_accuracy')])
    assert np.allclose(got[:, keys.index('train/lr')], gold[:, keys.index('train/lr')], rtol=1e-6, atol=0)


def test_opt125m_teacher_forced_all_64_steps():
    """BASELINE.json 'loss curves matching reference to 1e-4', made well posed at EVERY step of configs[0]: step k of
    the native fp32 path starts from the teacher's weights and Adam moments of step k and must reproduce the teacher's loss_k, gradient norm and
    parameter update.  No trajectory divergence enters: both sides evaluate the same function at the same point.  Reference:
    trainers/text_to_text/dpo.py:205-237 (train_step), supervised_trainer.py:234-257 (optimizer / schedule).

    Chain of evidence: (1) oracle/teacher.py on the CPU == the UNMODIFIED reference trainer, bit for bit, teacher-forced on the reference's own
    states at all 64 steps (build container; tests/golden/opt125m_teacher.npz, re-checked by tests/test_oracle_golden.py); (2) the same teacher
    code executed by torch on the GPU (HF modules, fp32, eager attention) drives the 64 steps here -- the host cores of the GPU box need 3.7 s per
    teacher step, the GPU 0.05 s -- and is tied to the CPU teacher at steps 0 / 21 / 42 / 63 on the same states; (3) the native HIP path against
    that teacher at every step."""
    from oracle.synthetic import opt125m_config1
    from oracle.teacher import Teacher, fingerprint
    from align_anything_amd import configs
    from align_anything_amd.trainers.dpo import DPOTrainer
    z = load_golden('opt125m_teacher.npz')
    oc, policy, refm, batches = opt125m_config1()
    steps = len(batches)
    cfg = configs.from_hf_config(oc)
    sd0 = {k: v.detach().clone() for k, v in policy.state_dict().items()}
    teacher = Teacher(cfg, refm.state_dict(), oc.pad_token_id, steps, hf_config=oc, device=dev())
    cpu_teacher = Teacher(cfg, refm.state_dict(), oc.pad_token_id, steps, hf_config=oc)
    cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-6, 'lr_warmup_ratio': 0.03, 'lr_scheduler_type': 'cosine', 'weight_decay': 0.05,
                           'adam_betas': [0.9, 0.95], 'total_training_steps': steps, 'compute_dtype': 'fp32'},
            'model_cfgs': {'pad_token_id': oc.pad_token_id}}
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, policy_state=sd0, reference_state=refm.state_dict(), device='cuda:0')
    eng, store = tr.model, tr.policy.store
    names = Teacher.names(sd0)
    index = {str(n): torch.from_numpy(i).to(dev()) for n, i in zip(z['fp_names'], z['fp_index'])}
    w = {k: v.to(dev()) for k, v in sd0.items()}
    m, v = Teacher.zeros_like(w), Teacher.zeros_like(w)
    keys = ('loss', 'gnorm', 'upd_mat', 'upd_all', 'frac', 'maxd', 'ref_loss', 'fp', 'cpu_loss', 'cpu_upd', 'm_err', 'v_err', 'cpu_m')
    lines, worst = [], dict.fromkeys(keys, 0.0)
    worst_name = ''
    for k, b in enumerate(batches):
        ti, w2, m2, v2 = teacher.step(w, m, v, k, b)
        if k in (0, 21, 42, steps - 1):     # the CPU teacher (== the reference, bit for bit) on the same state: loss and update of the GPU teacher
            tc, wc, mc, _ = cpu_teacher.step({n: t.cpu() for n, t in w.items()}, {n: t.cpu() for n, t in m.items()}, {n: t.cpu() for n, t in v.items()}, k, b)
            worst['cpu_loss'] = max(worst['cpu_loss'], abs(tc['train/loss'] - ti['train/loss']))
            for n in names:
                if n.endswith(Teacher.NOISE_ONLY):
                    continue
                worst['cpu_m'] = max(worst['cpu_m'], float((m2[n].cpu().double() - mc[n].double()).norm() / mc[n].double().norm().clamp_min(1e-30)))
                if w[n].dim() != 2 or 'layer_norm' in n:
                    continue
                du = (wc[n] - w[n].cpu()).double()
                worst['cpu_upd'] = max(worst['cpu_upd'], float(((w2[n].cpu() - w[n].cpu()).double() - du).norm() / du.norm().clamp_min(1e-30)))
            lines.append(f'step {k:2d} CPU teacher loss {tc["train/loss"]:.7f} (reference run: {float(z["ref"][k, 0]):.7f} on ITS trajectory) vs GPU teacher {ti["train/loss"]:.7f}')
        # ---- teacher forcing: the native step starts from the teacher's state of step k
        eng.wait_optimizer()
        store.load_state_dict(w)
        store.load_opt_state(m, v)
        eng.global_steps = k
        info = tr.train_step({'input_ids': b['input_ids'].to(dev()), 'attention_mask': b['attention_mask'].to(dev()), 'meta_info': b['meta_info']})
        gn = eng.grad_norm()
        e_loss = abs(info['train/loss'] - ti['train/loss'])
        e_gn = abs(gn - ti['grad_norm']) / ti['grad_norm']
        assert abs(info['train/lr'] - ti['train/lr']) <= 1e-12 * max(1.0, ti['train/lr'])
        # ---- the update of step k, element by element on the device.  An fp32 weight of magnitude |w| moves in quanta of ulp(|w|) (1.9e-9 at
        # 0.02, 1.2e-7 at a LayerNorm weight of 1.0) against a step of <= lr = 1e-6, so "same update" means: within a few quanta, except for the
        # rare element whose Adam direction is ill-conditioned (|g| at the noise level: m/sqrt(v) flips) -- bounded by 2 lr whatever happens.
        # The relative L2 error of the update itself is reported, not asserted: Adam divides every gradient element by its own magnitude history,
        # so elements whose gradient sits at the rounding-noise level move by +-lr in a direction no two fp32 implementations agree on (the same
        # teacher code on CPU and GPU differs by 3e-2 there).  What IS well conditioned and asserted at every step: the first and second moments
        # after the step (linear / quadratic in the clipped gradient: they carry the gradients, the clip coefficient and the moment update), the
        # element-wise bounds on the weights, the loss and the gradient norm.
        upd_mat, upd_all, frac_w, maxd, m_err, v_err = 0.0, 0.0, 0.0, 0.0, 0.0, 0.0
        for n in names:
            before, want, got = w[n], w2[n], store.view(n)
            if not n.endswith(Teacher.NOISE_ONLY):
                _, m_nat, v_nat = store.opt_state_views(n)
                m_err = max(m_err, float((m_nat.double() - m2[n].double()).norm() / m2[n].double().norm().clamp_min(1e-30)))
                v_err = max(v_err, float((v_nat.double() - v2[n].double()).norm() / v2[n].double().norm().clamp_min(1e-30)))
            d = (got - want).abs()
            quant = 4.0 * torch.finfo(torch.float32).eps * want.abs() + 2e-2 * ti['lr_used']
            maxd = max(maxd, float(d.max()) / max(ti['lr_used'], 1e-30))
            if n.endswith(Teacher.NOISE_ONLY):        # key biases: the true gradient is identically zero, the update is normalised rounding noise
                continue
            frac_w = max(frac_w, float((d > quant).float().mean()))
            du = (want - before).double()
            e = float(((got - before).double() - du).norm() / du.norm().clamp_min(1e-30))
            if e > upd_all:
                upd_all = e
                if e > worst['upd_all']:
                    worst_name = f'{n} at step {k}'
            if want.dim() == 2 and 'layer_norm' not in n:
                upd_mat = max(upd_mat, e)
        ref_dev = abs(ti['train/loss'] - float(z['ref'][k, 0]))
        lines.append(f'step {k:2d} loss native {info["train/loss"]:.7f} teacher {ti["train/loss"]:.7f} |diff| {e_loss:.1e}  gnorm rel {e_gn:.1e}  moments after the step rel-L2 (worst tensor): m '
                     f'{m_err:.1e} v {v_err:.1e}; update rel-L2: matrices {upd_mat:.1e}, any tensor {upd_all:.1e}; elements off by > 4 ulp + 2% lr: {frac_w:.1e}, max |dw| / lr {maxd:.2f}   '
                     f'(free-running teacher vs reference curve {ref_dev:.1e})')
        for key, val in (('loss', e_loss), ('gnorm', e_gn), ('upd_mat', upd_mat), ('upd_all', upd_all), ('frac', frac_w), ('maxd', maxd), ('ref_loss', ref_dev), ('m_err', m_err), ('v_err', v_err)):
            worst[key] = max(worst[key], val)
        w, m, v = w2, m2, v2
        worst['fp'] = max(worst['fp'], float((fingerprint(w, index).cpu() - torch.from_numpy(z['fingerprint'][k + 1])).abs().max()))
    lines.append(f'max over {steps} teacher-forced steps: |loss diff| {worst["loss"]:.2e} (target 1e-4), grad-norm rel {worst["gnorm"]:.2e}, Adam moments rel-L2 m {worst["m_err"]:.2e} '
                 f'v {worst["v_err"]:.2e}, update rel-L2 matrices '
                 f'{worst["upd_mat"]:.2e} / any tensor {worst["upd_all"]:.2e} ({worst_name}), fraction of elements beyond 4 ulp + 2 % lr {worst["frac"]:.2e}, max |dw| / lr {worst["maxd"]:.2f}')
    lines.append(f'GPU teacher vs CPU teacher on the same states (steps 0 / 21 / 42 / {steps - 1}): max |loss| {worst["cpu_loss"]:.2e}, first moment rel-L2 {worst["cpu_m"]:.2e}, '
                 f'update rel-L2 (matrices) {worst["cpu_upd"]:.2e}')
    lines.append(f'CPU teacher pinned to the reference in the build container (teacher-forced on the reference\'s own states, all {steps} steps): max |loss| '
                 f'{z["teacher_dev"][:, 0].max():.2e}, rel grad-norm {z["teacher_dev"][:, 1].max():.2e}, rel update {z["teacher_dev"][:, 2].max():.2e}; the oracle\'s own model '
                 f'port on the same states: max |loss| {z["teacher_dev"][:, 4].max():.2e}')
    lines.append(f'free-running GPU teacher vs the committed reference run: max |loss| {worst["ref_loss"]:.2e}, max |weight fingerprint diff| {worst["fp"]:.2e}')
    dump('parity_fp32_opt125m_teacher_forced.txt', '\n'.join(lines) + '\n')
    assert worst['cpu_loss'] < 5e-5 and worst['cpu_m'] < 2e-2, worst
    assert worst['loss'] < 1e-4, worst
    assert worst['gnorm'] < 1e-3, worst
    assert worst['m_err'] < 2e-2 and worst['v_err'] < 4e-2, worst
    assert worst['maxd'] <= 2.05, worst                       # no element moves further from the teacher than a flipped Adam direction can take it
    assert worst['frac'] < 2e-2, worst
    assert worst['ref_loss'] < 2e-3 and worst['fp'] < 64 * 2e-6, worst


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
@pytest.mark.parametrize('case', [(3, 150, 4, 4, 64, [150, 97, 1]), (2, 64, 2, 2, 128, [33, 64]), (1, 200, 2, 1, 64, [130])])
def test_attention_right_padding_kv_len(case, dtype):
    """Encoder attention over right-padded sequences (Whisper / Qwen2-Audio tower): keys >= kv_len[n] are masked
    (hf:models/qwen2_audio/modeling_qwen2_audio.py:686-712); forward and backward, both element types."""
    from align_anything_amd import ops
    N, T, H, Hkv, hd, lens = case
    dt = torch.float32 if dtype == 'fp32' else torch.bfloat16
    scale = hd ** -0.5
    qkv = rnd(N * T, (H + 2 * Hkv) * hd, seed=21).to(dt)
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:(H + Hkv) * hd], qkv[:, (H + Hkv) * hd:]
    do = rnd(N * T, H * hd, seed=22).to(dt)
    kvl = torch.tensor(lens, dtype=torch.int32, device=dev())
    idx = torch.arange(T, device=dev())
    valid = (idx[None, :] < kvl[:, None].long())                                    # [N, T]
    qf = q.double().view(N, T, H, hd).transpose(1, 2).detach().requires_grad_(True)
    kf = k.double().view(N, T, Hkv, hd).transpose(1, 2).detach().requires_grad_(True)
    vf = v.double().view(N, T, Hkv, hd).transpose(1, 2).detach().requires_grad_(True)
    s = (qf @ kf.repeat_interleave(H // Hkv, 1).transpose(-1, -2)) * scale
    p = torch.softmax(s.masked_fill(~valid[:, None, None, :], float('-inf')), -1)
    o_ref = p @ vf.repeat_interleave(H // Hkv, 1)
    dof = do.double().view(N, T, H, hd).transpose(1, 2) * valid[:, None, :, None]   # no loss gradient reaches pad frames
    (o_ref * dof).sum().backward()
    back = lambda t, h: t.transpose(1, 2).reshape(N * T, h * hd)
    o, lse = ops.attn_fwd(q, k, v, N, T, H, Hkv, hd, False, scale, kv_len=kvl)
    dqkv = torch.zeros_like(qkv)
    dq, dk, dv = dqkv[:, :H * hd], dqkv[:, H * hd:(H + Hkv) * hd], dqkv[:, (H + Hkv) * hd:]
    do_m = (do * valid.reshape(-1, 1).to(dt)).contiguous()
    ops.attn_bwd(q, k, v, o, do_m, lse, dq, dk, dv, N, T, H, Hkv, hd, False, scale, kv_len=kvl)
    vm = valid.reshape(-1, 1)
    tol = 1e-5 if dtype == 'fp32' else 3e-2
    assert rel_err(o.double() * vm, back(o_ref.detach(), H) * vm) < tol
    assert rel_err(dq.double() * vm, back(qf.grad, H) * vm) < 3 * tol
    assert rel_err(dk.double(), back(kf.grad, Hkv)) < 3 * tol and rel_err(dv.double(), back(vf.grad, Hkv)) < 3 * tol
    assert float((dk.float() * (~vm)).abs().max()) == 0.0 and float((dv.float() * (~vm)).abs().max()) == 0.0   # masked keys get no gradient
