"""GPU: parity AT THE BENCHMARKED GEOMETRY (BASELINE.json configs[1]: LLaVA-1.5-7B, h=4096, 32 heads x 128, ffn 11008,
V=32064, T=2048, 4 pairs/GPU = 16384 tokens), i.e. the kernel configurations `bench.py` actually times:

* every hot GEMM shape of the step in its layout (forward NT, dX NN, dW TN) so the tile orders that only large shapes
  select (column-grouped 0x104, GM=3) and the 64-tile-row grids execute under test -- against an fp32 matmul on a row
  sample that touches every tile row, plus a whole-output linearity check (column sums of C == (1^T A) B), which is
  independent of the sample;
* attention forward / backward at N=8, H=32, hd=128, T=2048, causal, one left-padded row (the XCD-balanced
  heaviest-first grid of 16 q-blocks x 32 heads x 8 rows) against the fp32 softmax reference on a head subset;
* one full 4096-wide decoder layer + final norm + lm_head + log-prob gather + DPO loss + backward at T=2048 through the
  trainer, against the CPU oracle (oracle/models.py::llama_decoder, oracle/rl_math.py) -- the path
  align_anything/trainers/text_image_to_text/dpo.py:85-166 executes per step.
"""
import math

import pytest
import torch

from tests.gpu_util import assert_close, dev, dump, randn_bf16
from tests.test_attention_gpu import ref_attention
from tests.util import rel_err

pytestmark = pytest.mark.gpu

H7B, F7B, V7B, TOK = 4096, 11008, 32064, 16384     # hidden, ffn, vocab, tokens per step (8 rows x 2048)
RESP_ROWS = 4096                                    # response-window rows of the step: 8 x (512 - 1) padded to 64 -> 4096

# (name, layout, M, N, K) exactly as modeling.LlamaStack / LMHead issue them (Linear.fwd / dx / dw)
HOT_GEMMS = [
    ('qkv.fwd', 'nt', TOK, 3 * H7B, H7B), ('o.fwd', 'nt', TOK, H7B, H7B),
    ('gate_up.fwd', 'nt', TOK, 2 * F7B, H7B), ('down.fwd', 'nt', TOK, H7B, F7B),
    ('qkv.dx', 'nn', TOK, H7B, 3 * H7B), ('o.dx', 'nn', TOK, H7B, H7B),
    ('gate_up.dx', 'nn', TOK, H7B, 2 * F7B), ('down.dx', 'nn', TOK, F7B, H7B),
    ('qkv.dw', 'tn', 3 * H7B, H7B, TOK), ('o.dw', 'tn', H7B, H7B, TOK),
    ('gate_up.dw', 'tn', 2 * F7B, H7B, TOK), ('down.dw', 'tn', H7B, F7B, TOK),
    ('lm_head.fwd', 'nt', RESP_ROWS, V7B, H7B), ('lm_head.dx', 'nn', RESP_ROWS, H7B, V7B),
    ('lm_head.dw', 'tn', V7B, H7B, RESP_ROWS),
]


def _rand(rows, cols, seed, scale=1.0):
    g = torch.Generator(device=dev()).manual_seed(seed)
    return (torch.randn((rows, cols), generator=g, device=dev()) * scale).to(torch.bfloat16)


@pytest.mark.parametrize('case', HOT_GEMMS, ids=[c[0] for c in HOT_GEMMS])
def test_hot_gemm_shapes_of_the_benchmarked_step(case):
    check_gemm_case(case)


def check_gemm_case(case):
    """One production GEMM launch (name, layout, M, N, K) against an fp32 matmul on a row sample that touches every tile row, plus the
    whole-output column-sum identity.  Shared with tests/test_secondary_geometry_gpu.py (the 7B shapes of configs[2]-[4])."""
    from align_anything_amd import ops
    name, layout, M, N, K = case
    a_t, b_n = layout == 'tn', layout in ('nn', 'tn')
    a = _rand(K, M, 1) if a_t else _rand(M, K, 1)
    b = _rand(K, N, 2, 0.05) if b_n else _rand(N, K, 2, 0.05)
    out = ops.gemm(a, b, a_t=a_t, b_n=b_n)
    torch.cuda.synchronize()
    assert out.shape == (M, N) and torch.isfinite(out.float()).all()
    # (1) a row sample that hits every 256-row tile (and every 128-row tile) at a varying offset, all columns
    g = torch.Generator(device='cpu').manual_seed(3)
    base = torch.arange(0, M, 128)
    rows = (base + torch.randint(0, 128, (base.numel(),), generator=g)).clamp_(max=M - 1).to(dev())
    A_rows = (a[:, rows].t() if a_t else a[rows]).float()
    Bf = b.float() if b_n else b.float().t()
    ref = A_rows @ Bf
    scale = float(ref.abs().mean())
    assert_close(out[rows], ref, rtol=1e-2, atol=1e-2 * scale, what=f'{name} {layout} {M}x{N}x{K} rows')
    # (2) linearity over ALL rows: sum_m C[m, :] == (sum_m A[m, :]) B   (fp64 reference; bf16 output rounding averages out)
    colsum = out.double().sum(0)
    a_sum = (a.double().sum(1) if a_t else a.double().sum(0))
    want = a_sum @ Bf.double()
    tol = 4e-3 * math.sqrt(M) * scale * 8 + 1e-2 * float(want.abs().mean())
    err = float((colsum - want).abs().max())
    assert err < tol, f'{name}: column-sum mismatch {err} (tol {tol})'


def test_attention_at_the_benchmarked_geometry():
    """N=8 rows (4 pairs), H=32, hd=128, T=2048, causal; row 5 left-padded by 333 tokens, row 2 by 64."""
    from align_anything_amd import ops
    N, T, H, hd = 8, 2048, 32, 128
    scale = hd ** -0.5
    qkv = _rand(N * T, 3 * H * hd, 11, 0.7)
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:2 * H * hd], qkv[:, 2 * H * hd:]
    do = _rand(N * T, H * hd, 12)
    start = torch.tensor([0, 0, 64, 0, 0, 333, 0, 0], dtype=torch.int32, device=dev())
    idx = torch.arange(T, device=dev())
    valid = (idx[None, :] >= start[:, None].long()).reshape(N * T)
    do = do * valid[:, None].to(do.dtype)
    o, lse = ops.attn_fwd(q, k, v, N, T, H, H, hd, True, scale, start)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[:, :H * hd], dqkv[:, H * hd:2 * H * hd], dqkv[:, 2 * H * hd:]
    ops.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, N, T, H, H, hd, True, scale, start)
    torch.cuda.synchronize()
    assert torch.isfinite(o.float()).all() and torch.isfinite(dqkv.float()).all()
    vm = valid[:, None].float()
    assert float((o.float() * (1 - vm)).abs().max()) == 0.0, 'pad query rows must be exactly 0'
    rep = []
    for head in (0, 13, 31):
        sl = slice(head * hd, (head + 1) * hd)
        ro, rdq, rdk, rdv, _ = ref_attention(q[:, sl], k[:, sl], v[:, sl], do[:, sl], N, T, 1, 1, hd, True, scale, start)
        assert_close(o[:, sl].float() * vm, ro * vm, rtol=2e-2, atol=2e-2, what=f'O head {head}')
        for nm, got, want in (('dQ', dq[:, sl].float() * vm, rdq * vm), ('dK', dk[:, sl], rdk), ('dV', dv[:, sl], rdv)):
            assert_close(got, want, rtol=3e-2, atol=2e-2 * max(float(want.abs().max()), 1e-3), what=f'{nm} head {head}')
            rep.append(f'head {head} {nm} rel_err {rel_err(got.float(), want):.5f}')
        rep.append(f'head {head} O rel_err {rel_err(o[:, sl].float() * vm, ro * vm):.5f}')
    dump('parity_attention_T2048.txt', '\n'.join(rep) + '\n')


def test_full_width_decoder_layer_dpo_step_matches_oracle():
    """One Llama layer of the 7B geometry (h=4096, 32x128 heads, ffn 11008, V=32064) at T=2048, one pair, row 1 left
    padded: response log-probs, DPO loss and EVERY parameter gradient against the fp32 CPU oracle."""
    from align_anything_amd import configs
    from align_anything_amd.trainers.dpo import DPOTrainer
    from oracle import models as om
    from oracle import rl_math as orl
    cfg = configs.llama_cfg(H7B, F7B, 1, 32, 32, V7B, rms_eps=1e-5, max_position_embeddings=4096)
    Tn, R, pad_id, beta = 2048, 512, 0, 0.1
    g = torch.Generator(device='cpu').manual_seed(7)
    h, F, V = H7B, F7B, V7B
    p = 'model.layers.0.'
    shapes = {'model.embed_tokens.weight': (V, h), p + 'self_attn.q_proj.weight': (h, h), p + 'self_attn.k_proj.weight': (h, h),
              p + 'self_attn.v_proj.weight': (h, h), p + 'self_attn.o_proj.weight': (h, h), p + 'mlp.gate_proj.weight': (F, h),
              p + 'mlp.up_proj.weight': (F, h), p + 'mlp.down_proj.weight': (h, F), 'lm_head.weight': (V, h)}
    sd = {k: (torch.randn(s, generator=g) * 0.02).to(torch.bfloat16) for k, s in shapes.items()}
    for k in (p + 'input_layernorm.weight', p + 'post_attention_layernorm.weight', 'model.norm.weight'):
        sd[k] = (1.0 + 0.1 * torch.randn(h, generator=g)).to(torch.bfloat16)
    # reference model = a perturbed copy, so the DPO log-ratio (and hence d loss / d logp) is not degenerate
    sd_ref = {k: (v.float() + 0.002 * torch.randn(v.shape, generator=g)).to(torch.bfloat16) if v.dim() == 2 else v.clone() for k, v in sd.items()}
    ids = torch.randint(3, V, (2, Tn), generator=g)
    am = torch.ones(2, Tn, dtype=torch.long)
    ids[1, :100] = pad_id
    am[1, :100] = 0
    ids[1, :Tn - R] = torch.where(am[1, :Tn - R].bool(), ids[0, :Tn - R], ids[1, :Tn - R])
    lens = [R, R - 37]
    cfgs = {'train_cfgs': {'scale_coeff': beta, 'learning_rate': 1e-6, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant'},
            'model_cfgs': {'pad_token_id': pad_id}}
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, policy_state=sd, reference_state=sd_ref, device='cuda:0')
    batch = {'input_ids': ids.to(dev()), 'attention_mask': am.to(dev()), 'meta_info': {'response_lens': lens}}
    lp = tr.compute_log_probs(tr.model, batch).cpu()
    ld = tr.loss(batch)
    tr.model.backward(ld['loss'])
    torch.cuda.synchronize()

    # ---- oracle (fp32, CPU): decoder -> lm_head on all positions -> the reference's compute_log_probs -> DPO loss
    osd = {k: v.float().requires_grad_(True) for k, v in sd.items()}
    with torch.no_grad():
        rsd = {k: v.float() for k, v in sd_ref.items()}
        ref_lp = orl.compute_log_probs(om.llama_logits(rsd, cfg, ids, am), ids, lens, pad_id)
    want_lp = orl.compute_log_probs(om.llama_logits(osd, cfg, ids, am), ids, lens, pad_id)
    want = orl.dpo_loss(want_lp, ref_lp, beta)
    want['loss'].backward()
    rep = [f'loss native {float(ld["loss"]):.6f} oracle {float(want["loss"]):.6f}']
    assert torch.equal(lp == 0, want_lp.detach() == 0), 'response-window layout differs'
    assert_close(lp, want_lp.detach(), rtol=2e-2, atol=5e-2, what='response log-probs at h=4096/T=2048')
    assert abs(float(ld['loss']) - float(want['loss'])) < 1e-2
    st = tr.policy.store
    worst = 0.0
    for k, v in osd.items():
        got = st.grad_view(k).float().cpu().reshape(v.shape)
        e = rel_err(got, v.grad)
        rep.append(f'grad {k}: rel_err {e:.4f} |want| {float(v.grad.norm()):.3e}')
        worst = max(worst, e)
    dump('parity_layer_h4096_T2048.txt', '\n'.join(rep) + f'\nworst grad rel err {worst:.4f}\n')
    assert worst < 6e-2, rep
