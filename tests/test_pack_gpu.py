"""GPU: shared-prompt packing (trainers.common.build_pack_plan, train_cfgs.share_prompt_prefix) against the unpacked computation of the same trainer.
The reference (trainers/text_image_to_text/dpo.py:85-105) runs every row of the [2B, T] batch through the model; the packed path runs a pair's common
prefix once.  Claims: (1) with equal left padding the packed forward is BIT-IDENTICAL (row-wise kernels never see other rows, attention sees the
reference layout); (2) with ragged pairs the rejected row is evaluated in the chosen row's rotary frame -- equal up to the rounding of the rotary tables;
(3) gradients agree up to the rounding of the sum of a shared row's two gradient copies in the activation dtype.  The same switch is held against the
UNMODIFIED reference trainer at full depth in tests/test_secondary_geometry_gpu.py."""
import numpy as np
import pytest
import torch

from tests.gpu_util import dev, dump
from tests.util import load_golden, rel_err, state_dict_from_golden, tiny_llava_cfg

pytestmark = pytest.mark.gpu


def _pair_batch(B, T, prompt_lens, resp_c, resp_r, seed=0, image_tokens=4):
    g = torch.Generator().manual_seed(seed)
    N = 2 * B
    ids = torch.full((N, T), 301, dtype=torch.long)
    mask = torch.zeros((N, T), dtype=torch.long)
    for i in range(B):
        prompt = torch.cat([torch.tensor([1]), torch.full((image_tokens,), 300, dtype=torch.long), torch.randint(3, 299, (prompt_lens[i] - 1 - image_tokens,), generator=g)])
        for row, R in ((i, resp_c[i]), (B + i, resp_r[i])):
            seq = torch.cat([prompt, torch.randint(3, 299, (R,), generator=g)])
            ids[row, T - len(seq):] = seq
            mask[row, T - len(seq):] = 1
    pix = torch.randn(B, 3, 28, 28, generator=g)
    return {'input_ids': ids.to(dev()), 'attention_mask': mask.to(dev()), 'pixel_values': torch.cat([pix, pix], 0).to(dev()),
            'meta_info': {'response_lens': list(resp_c) + list(resp_r)}}


def _run(dtype, share, batch_args):
    from align_anything_amd.trainers.dpo import DPOTrainer
    z = load_golden('llava_tiny_dpo.npz')
    cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-4, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant', 'compute_dtype': dtype,
                           'share_prompt_prefix': share}, 'model_cfgs': {'pad_token_id': 301}}
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=tiny_llava_cfg(), policy_state=state_dict_from_golden(z, 'w.', torch.bfloat16),
                    reference_state=state_dict_from_golden(z, 'r.', torch.bfloat16), device='cuda:0')
    b = _pair_batch(*batch_args)
    # every attention output buffer starts as NaN: the rows of a skipped query block are never written, and nothing may read them (tests/test_tail_gpu.py)
    from align_anything_amd import ops
    real = ops.attn_fwd

    def attn_fwd(q, k, v, N, T, H, Hkv, hd, *a, **kw):
        if kw.get('out') is None:
            kw['out'] = torch.full((q.shape[0], H * hd), float('nan'), dtype=q.dtype, device=q.device)
        return real(q, k, v, N, T, H, Hkv, hd, *a, **kw)
    ops.attn_fwd = attn_fwd
    try:
        lp = tr.compute_log_probs(tr.model, b)
        rlp = tr.compute_log_probs(tr.reference_model, b)
        ld = tr.loss(b)
        tr.model.backward(ld['loss'])
        torch.cuda.synchronize()
    finally:
        ops.attn_fwd = real
    st = tr.policy.store
    grads = {n: st.grad_view(n).float().clone() for n in st.hf_names() if st.grad_view(n) is not None}
    assert all(bool(torch.isfinite(g).all()) for g in grads.values()) and bool(torch.isfinite(lp.float()).all())
    plan = b.get('_pack')
    return lp.float().cpu(), rlp.float().cpu(), float(ld['loss']), grads, plan


@pytest.mark.parametrize('dtype', ['bf16', 'fp32'])
def test_packing_with_equal_padding_is_bit_identical_in_the_forward_pass(dtype):
    args = (3, 256, (150, 100, 192), (64, 100, 40), (64, 100, 40), 1)          # chosen / rejected responses of equal length: the same left padding in both rows
    lp0, rlp0, loss0, g0, plan0 = _run(dtype, False, args)
    lp1, rlp1, loss1, g1, plan1 = _run(dtype, True, args)
    assert plan0 is None and plan1 is not None and plan1['prefix_lens'] == [150, 100, 192]
    assert plan1['rows'] == 2 * (214 + 200 + 232) - (150 + 100 + 192)
    assert torch.equal(lp0, lp1) and torch.equal(rlp0, rlp1), float((lp0 - lp1).abs().max())
    assert loss0 == loss1
    worst = max(rel_err(g1[n], g0[n]) for n in g0 if float(g0[n].norm()) > 1e-6)
    dump(f'parity_pack_equal_padding_{dtype}.txt', f'forward bit-identical; worst gradient rel_err packed vs unpacked {worst:.3e} over {len(g0)} tensors\n')
    assert worst < (2e-2 if dtype == 'bf16' else 2e-6), worst


@pytest.mark.parametrize('dtype', ['bf16', 'fp32'])
def test_packing_of_ragged_pairs_matches_the_unpacked_computation(dtype):
    args = (3, 256, (150, 100, 80), (64, 30, 120), (20, 100, 96), 2)           # different response lengths: different left padding, the rejected row moves into the chosen row's rotary frame
    lp0, rlp0, loss0, g0, _ = _run(dtype, False, args)
    lp1, rlp1, loss1, g1, plan = _run(dtype, True, args)
    assert plan is not None and plan['prefix_lens'] == [150, 100, 80]
    assert torch.equal(lp0 == 0, lp1 == 0)
    e_lp = float((lp0 - lp1).abs().max())
    worst = max(rel_err(g1[n], g0[n]) for n in g0 if float(g0[n].norm()) > 1e-6)
    dump(f'parity_pack_ragged_{dtype}.txt', f'per-token log-probs packed vs unpacked {e_lp:.3e}; loss {abs(loss0 - loss1):.3e}; worst gradient rel_err {worst:.3e}\n')
    if dtype == 'fp32':
        assert e_lp < 2e-5 and abs(loss0 - loss1) < 1e-6 and worst < 1e-4, (e_lp, loss0 - loss1, worst)
    else:
        assert e_lp < 6e-2 and abs(loss0 - loss1) < 5e-3 and worst < 3e-2, (e_lp, loss0 - loss1, worst)


def test_a_pair_without_a_common_prefix_keeps_the_reference_layout():
    from align_anything_amd.trainers.common import build_pack_plan, build_window
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(3, 299, (4, 128), generator=g).to(dev())
    w = build_window(ids, [16] * 4, 301)
    assert build_pack_plan(ids, torch.ones_like(ids), w, {'response_lens': [16] * 4}) is None


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_gather2_add_equals_two_gathers_and_an_add(dtype):
    from align_anything_amd import ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn(300, 256, generator=g).to(dtype).to(dev())
    a = torch.randint(-1, 300, (257,), generator=g).to(torch.int32).to(dev())
    b = torch.randint(-1, 300, (257,), generator=g).to(torch.int32).to(dev())
    b[::3] = -1
    got = ops.gather2_add(x, a, b)
    want = ops.add(ops.moe_gather(x, a), ops.moe_gather(x, b))
    assert torch.equal(got, want)
    assert ops.gather2_add(x, a[:0], b[:0]).shape == (0, 256)


def test_packing_on_the_text_llama_trainer_gqa_ragged():
    """The text-to-text DPO path (trainers/text_to_text/dpo.py on a Llama-family decoder, GQA 4 / 2): packed against unpacked on ragged pairs, fp32 twin."""
    from align_anything_amd import configs
    from align_anything_amd.trainers.dpo import DPOTrainer
    cfg = configs.llama_cfg(256, 512, 2, 4, 2, 320, rms_eps=1e-5, max_position_embeddings=256)
    out = {}
    for share in (False, True):
        cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-4, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant', 'compute_dtype': 'fp32',
                               'share_prompt_prefix': share}, 'model_cfgs': {'pad_token_id': 301}}
        tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, device='cuda:0')
        g = torch.Generator().manual_seed(9)
        for n in tr.policy.store.hf_names():
            v = tr.policy.store.view(n)
            w = torch.randn(tuple(v.shape), generator=g) * 0.05 + (1.0 if 'norm' in n else 0.0)
            v.copy_(w.to(dev())); tr.reference.store.view(n).copy_((w * 1.01).to(dev()))
        for gname in tr.policy.store.master:
            if tr.policy.store.master[gname] is not tr.policy.store.flat[gname]:
                tr.policy.store.master[gname].copy_(tr.policy.store.flat[gname])
        b = _pair_batch(2, 192, (100, 70), (30, 60), (80, 20), seed=5, image_tokens=0)
        b.pop('pixel_values')
        lp = tr.compute_log_probs(tr.model, b).float().cpu()
        ld = tr.loss(b)
        tr.model.backward(ld['loss'])
        torch.cuda.synchronize()
        st = tr.policy.store
        out[share] = (lp, float(ld['loss']), {n: st.grad_view(n).float().clone() for n in st.hf_names() if st.grad_view(n) is not None}, b.get('_pack'))
    (lp0, l0, g0, p0), (lp1, l1, g1, p1) = out[False], out[True]
    assert p0 is None and p1 is not None and p1['prefix_lens'] == [100, 70]
    worst = max(rel_err(g1[n], g0[n]) for n in g0 if float(g0[n].norm()) > 1e-6)
    assert float((lp0 - lp1).abs().max()) < 2e-5 and abs(l0 - l1) < 1e-6 and worst < 1e-4, (float((lp0 - lp1).abs().max()), l0 - l1, worst)


def test_packing_is_inherited_by_the_sibling_preference_trainers_and_refuses_unshared_images():
    """SimPOTrainer overrides `loss` only (trainers/text_to_text/simpo.py:41-108): with the switch on its loss on equal-padding pairs equals the unpacked one bit
    for bit.  And rows that carry DIFFERENT images are never packed, although their image placeholder ids look alike."""
    from align_anything_amd.trainers.dpo import DPOTrainer
    from align_anything_amd.trainers.pref import SimPOTrainer
    z = load_golden('llava_tiny_dpo.npz')
    args = (2, 256, (150, 100), (64, 100), (64, 100), 7)
    losses = []
    for share in (False, True):
        cfgs = {'train_cfgs': {'scale_coeff': 2.5, 'gamma': 1.4, 'learning_rate': 1e-4, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant',
                               'share_prompt_prefix': share}, 'model_cfgs': {'pad_token_id': 301}}
        tr = SimPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=tiny_llava_cfg(), policy_state=state_dict_from_golden(z, 'w.', torch.bfloat16), device='cuda:0')
        b = _pair_batch(*args)
        losses.append(float(tr.loss(b)['loss']))
        assert (b.get('_pack') is not None) == share
    assert losses[0] == losses[1]
    cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-4, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant', 'share_prompt_prefix': True},
            'model_cfgs': {'pad_token_id': 301}}
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=tiny_llava_cfg(), policy_state=state_dict_from_golden(z, 'w.', torch.bfloat16),
                    reference_state=state_dict_from_golden(z, 'r.', torch.bfloat16), device='cuda:0')
    b = _pair_batch(*args)
    b['pixel_values'][2:] += 1.0                       # the rejected rows now carry other images
    assert tr._pack_plan(b) is None


def test_packing_on_the_qwen2audio_trainer_runs_the_tower_once_per_pair():
    """The text-audio DPO path (trainers/text_audio_to_text/dpo.py; Qwen2-Audio: trainable tower, q/k/v biases in the decoder): the clip of a pair is stacked
    twice by the collator (datasets/text_audio_to_text/preference.py:178-229); packed, its audio tokens appear once, the tower runs on the first half only, and
    the tower's gradient (chosen + rejected contributions through ONE set of rows) equals the unpacked one.  Ragged pairs, clips of 64 and 37 frames, fp32 twin."""
    from align_anything_amd.trainers.dpo import DPOTrainer
    from tests.test_qwen2audio_gpu import _sd
    from tests.util import tiny_qwen2audio_cfg
    z = load_golden('qwen2audio_tiny_dpo.npz')
    B, T = 2, 224
    g = torch.Generator().manual_seed(3)
    ids = torch.full((2 * B, T), 304, dtype=torch.long); mask = torch.zeros((2 * B, T), dtype=torch.long)
    frames, ntok, plen, rc, rr = (64, 37), (16, 9), (120, 90), (40, 70), (90, 25)
    for i in range(B):
        prompt = torch.cat([torch.tensor([1]), torch.full((ntok[i],), 300, dtype=torch.long), torch.randint(3, 299, (plen[i] - 1 - ntok[i],), generator=g)])
        for row, R in ((i, rc[i]), (B + i, rr[i])):
            seq = torch.cat([prompt, torch.randint(3, 299, (R,), generator=g)])
            ids[row, T - len(seq):] = seq; mask[row, T - len(seq):] = 1
    feat = torch.randn(B, 64, 64, generator=g)
    fmask = torch.zeros(B, 64, dtype=torch.long)
    for i in range(B):
        fmask[i, :frames[i]] = 1
    out = {}
    for share in (False, True):
        cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant', 'weight_decay': 0.0,
                               'compute_dtype': 'fp32', 'share_prompt_prefix': share}, 'model_cfgs': {'pad_token_id': 304}}
        tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=tiny_qwen2audio_cfg(), policy_state=_sd(z, 'w.', torch.float32),
                        reference_state=_sd(z, 'r.', torch.float32), device='cuda:0')
        b = {'input_ids': ids.to(dev()), 'attention_mask': mask.to(dev()), 'input_features': torch.cat([feat, feat], 0).to(dev()),
             'feature_attention_mask': torch.cat([fmask, fmask], 0).to(dev()), 'meta_info': {'response_lens': list(rc) + list(rr)}}
        lp = tr.compute_log_probs(tr.model, b).float().cpu()
        ld = tr.loss(b)
        tr.model.backward(ld['loss'])
        torch.cuda.synchronize()
        st = tr.policy.store
        out[share] = (lp, float(ld['loss']), {n: st.grad_view(n).float().clone() for n in st.hf_names() if st.grad_view(n) is not None}, b.get('_pack'))
        if share:
            b2 = dict(b); b2.pop('_pack', None)
            b2['input_features'] = b['input_features'].clone(); b2['input_features'][B:] += 1.0       # other clips on the rejected rows: never packed
            assert tr._pack_plan(b2) is None
    (lp0, l0, g0, p0), (lp1, l1, g1, p1) = out[False], out[True]
    assert p0 is None and p1 is not None and p1['prefix_lens'] == [120, 90]
    groups = {n.split('.')[1] if n.startswith('model.') else n.split('.')[0] for n in g0}
    worst = max(rel_err(g1[n], g0[n]) for n in g0 if float(g0[n].norm()) > 1e-6)
    tower = max(rel_err(g1[n], g0[n]) for n in g0 if 'audio_tower' in n and float(g0[n].norm()) > 1e-6)
    dump('parity_pack_qwen2audio.txt', f'packed vs unpacked, fp32 twin: max |dlogp| {float((lp0 - lp1).abs().max()):.2e}, loss {l0:.6f} / {l1:.6f}, worst gradient rel_err {worst:.2e} '
         f'(audio tower {tower:.2e}) over {len(g0)} tensors; prefix_lens {p1["prefix_lens"]}, packed rows {p1["rows"]} of {2 * B * T}\n')
    assert any('audio_tower' in n for n in g0), groups
    assert float((lp0 - lp1).abs().max()) < 2e-5 and abs(l0 - l1) < 1e-6 and worst < 1e-4, (float((lp0 - lp1).abs().max()), l0 - l1, worst)


@pytest.mark.parametrize('train_tower', [False, True])
def test_packing_on_the_qwen2vl_trainer_under_multimodal_rope(train_tower):
    """The Qwen2-VL DPO path (the configs[2] backbone; q/k/v biases, GQA, multimodal RoPE from the 3-D position index of the [N, T] layout): ragged pairs, images of
    6 and 4 merged tokens stacked twice (flattened patches + grids), packed against unpacked on the fp32 twin -- also with the vision tower training, whose
    gradient then arrives through ONE set of image rows per pair."""
    from tests.test_qwen2vl_gpu import _trainer
    z = load_golden('qwen2vl_tiny_dpo.npz')
    B, T = 2, 224
    g = torch.Generator().manual_seed(4)
    ids = torch.full((2 * B, T), 304, dtype=torch.long); mask = torch.zeros((2 * B, T), dtype=torch.long)
    grids, ntok, plen, rc, rr = [[1, 4, 6], [1, 4, 4]], (6, 4), (110, 84), (50, 72), (96, 31)
    for i in range(B):
        prompt = torch.cat([torch.tensor([1, 302]), torch.full((ntok[i],), 300, dtype=torch.long), torch.tensor([303]), torch.randint(3, 299, (plen[i] - 3 - ntok[i],), generator=g)])
        for row, R in ((i, rc[i]), (B + i, rr[i])):
            seq = torch.cat([prompt, torch.randint(3, 299, (R,), generator=g)])
            ids[row, T - len(seq):] = seq; mask[row, T - len(seq):] = 1
    pix = torch.randn(24 + 16, 1176, generator=g)
    out = {}
    for share in (False, True):
        tr = _trainer(z, 'fp32', train_tower)
        tr.share_prompt_prefix = share
        b = {'input_ids': ids.to(dev()), 'attention_mask': mask.to(dev()), 'pixel_values': torch.cat([pix, pix], 0).to(dev()),
             'image_grid_thw': torch.tensor(grids + grids), 'meta_info': {'response_lens': list(rc) + list(rr)}}
        lp = tr.compute_log_probs(tr.model, b).float().cpu()
        tr.policy.validate_batch()
        ld = tr.loss(b)
        tr.model.backward(ld['loss'])
        torch.cuda.synchronize()
        st = tr.policy.store
        out[share] = (lp, float(ld['loss']), {n: st.grad_view(n).float().clone() for n in st.hf_names() if st.grad_view(n) is not None}, b.get('_pack'))
        if share:
            b2 = {k: v for k, v in b.items() if not k.startswith('_')}
            b2['pixel_values'] = b['pixel_values'].clone(); b2['pixel_values'][40:] += 1.0            # other images on the rejected rows: never packed
            assert tr._pack_plan(b2) is None
    (lp0, l0, g0, p0), (lp1, l1, g1, p1) = out[False], out[True]
    assert p0 is None and p1 is not None and p1['prefix_lens'] == [110, 84]
    worst = max((rel_err(g1[n], g0[n]), n) for n in g0 if float(g0[n].norm()) > 1e-6)
    vis = [rel_err(g1[n], g0[n]) for n in g0 if n.startswith('model.visual.') and float(g0[n].norm()) > 1e-6]
    dump(f'parity_pack_qwen2vl{"_tower" if train_tower else ""}.txt', f'packed vs unpacked, fp32 twin{", vision tower training" if train_tower else ""}: max |dlogp| '
         f'{float((lp0 - lp1).abs().max()):.2e}, loss {l0:.6f} / {l1:.6f}, worst gradient rel_err {worst[0]:.2e} ({worst[1]}; visual.* {max(vis):.2e} over {len(vis)}) over {len(g0)} '
         f'tensors; prefix_lens {p1["prefix_lens"]}, packed rows {p1["rows"]} of {2 * B * T}\n')
    assert len(vis) > (10 if train_tower else 1)
    assert float((lp0 - lp1).abs().max()) < 2e-5 and abs(l0 - l1) < 1e-6 and worst[0] < 1e-4, (float((lp0 - lp1).abs().max()), l0 - l1, worst)


@pytest.mark.parametrize('ragged', [False, True])
def test_packing_on_the_qwen3moe_trainer_routes_the_shared_prefix_once(ragged):
    """The Qwen3-MoE DPO path (models/qwen3_moe.py; per-head q / k norms, router + top-k experts per token): a shared prefix row is normalised, routed and sent
    through its experts ONCE -- its routing is that of either copy (same hidden state).  Equal padding: log-probs bit-identical to the unpacked step; ragged
    pairs: one rotary frame per pair, compared on the fp32 twin (a flipped top-k choice would show as a gross difference)."""
    from tests.test_qwen3moe_gpu import _trainer
    z = load_golden('qwen3moe_tiny_dpo.npz')
    args = (2, 224, (120, 90), (40, 70), (90, 25), 6, 0) if ragged else (2, 224, (150, 100), (60, 110), (60, 110), 6, 0)
    out = {}
    for share in (False, True):
        tr = _trainer(z, 'fp32')
        tr.share_prompt_prefix = share
        tr.pad_token_id = 301
        b = _pair_batch(*args)
        b.pop('pixel_values')
        b['input_ids'] = b['input_ids'].clamp(max=int(z['input_ids'].max()))
        lp = tr.compute_log_probs(tr.model, b).float().cpu()
        ld = tr.loss(b)
        tr.model.backward(ld['loss'])
        torch.cuda.synchronize()
        st = tr.policy.store
        out[share] = (lp, float(ld['loss']), {n: st.grad_view(n).float().clone() for n in st.hf_names() if st.grad_view(n) is not None}, b.get('_pack'))
    (lp0, l0, g0, p0), (lp1, l1, g1, p1) = out[False], out[True]
    assert p0 is None and p1 is not None and p1['prefix_lens'] == ([120, 90] if ragged else [150, 100])
    worst = max((rel_err(g1[n], g0[n]), n) for n in g0 if float(g0[n].norm()) > 1e-6)
    dump(f'parity_pack_qwen3moe_{"ragged" if ragged else "equal_padding"}.txt', f'packed vs unpacked, fp32 twin: max |dlogp| {float((lp0 - lp1).abs().max()):.2e}, loss {l0:.6f} / {l1:.6f}, '
         f'worst gradient rel_err {worst[0]:.2e} ({worst[1]}) over {len(g0)} tensors; packed rows {p1["rows"]} of {4 * 224}\n')
    if not ragged:
        assert torch.equal(lp0, lp1) and l0 == l1
    assert float((lp0 - lp1).abs().max()) < 2e-5 and abs(l0 - l1) < 1e-6 and worst[0] < 1e-4, (float((lp0 - lp1).abs().max()), l0 - l1, worst)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_heavily_padded_pairs_without_a_common_prefix_are_packed_for_their_padding_alone(dtype):
    """Nothing to share (the two rows differ from their first tokens on) but more than 1/8 of the slots are left padding: the plan still takes the pad slots out
    of every row-wise kernel (text-to-text DPO on a Llama-family decoder).  Each row keeps its own positions, so the forward is bit-identical to the reference
    layout.  A MULTIMODAL pair without a shared prefix is never packed: its rows would each need their own copy of the image features."""
    from align_anything_amd import configs
    from align_anything_amd.trainers.dpo import DPOTrainer
    cfg = configs.llama_cfg(256, 512, 2, 4, 2, 320, rms_eps=1e-5, max_position_embeddings=256)
    out = {}
    for share in (False, True):
        cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-4, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant', 'compute_dtype': dtype,
                               'share_prompt_prefix': share}, 'model_cfgs': {'pad_token_id': 301}}
        tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, device='cuda:0')
        g = torch.Generator().manual_seed(9)
        for n in tr.policy.store.hf_names():
            v = tr.policy.store.view(n)
            w = torch.randn(tuple(v.shape), generator=g) * 0.05 + (1.0 if 'norm' in n else 0.0)
            v.copy_(w.to(dev())); tr.reference.store.view(n).copy_((w * 1.01).to(dev()))
        for gname in tr.policy.store.master:
            if tr.policy.store.master[gname] is not tr.policy.store.flat[gname]:
                tr.policy.store.master[gname].copy_(tr.policy.store.flat[gname])
        b = _pair_batch(2, 256, (90, 60), (40, 70), (80, 30), seed=13, image_tokens=0)
        b.pop('pixel_values')
        for r in (2, 3):                        # break the common prefix right after BOS
            first = int((b['attention_mask'][r] == 1).nonzero()[0])
            b['input_ids'][r, first + 1] = (b['input_ids'][r, first + 1] + 7) % 290 + 3
        lp = tr.compute_log_probs(tr.model, b).float().cpu()
        ld = tr.loss(b)
        tr.model.backward(ld['loss'])
        torch.cuda.synchronize()
        st = tr.policy.store
        out[share] = (lp, float(ld['loss']), {n: st.grad_view(n).float().clone() for n in st.hf_names() if st.grad_view(n) is not None}, b.get('_pack'))
    (lp0, l0, g0, p0), (lp1, l1, g1, p1) = out[False], out[True]
    assert p0 is None and p1 is not None and p1['prefix_lens'] == [0, 0] and p1['shared_rows'] == 0 and p1['rows'] < 0.6 * 4 * 256
    assert torch.equal(lp0, lp1) and l0 == l1
    worst = max(rel_err(g1[n], g0[n]) for n in g0 if float(g0[n].norm()) > 1e-6)
    assert worst < (1e-5 if dtype == 'fp32' else 4e-3), worst
    if dtype == 'fp32':
        z = load_golden('llava_tiny_dpo.npz')
        cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-4, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant', 'share_prompt_prefix': True},
                'model_cfgs': {'pad_token_id': 301}}
        tv = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=tiny_llava_cfg(), policy_state=state_dict_from_golden(z, 'w.', torch.bfloat16),
                        reference_state=state_dict_from_golden(z, 'r.', torch.bfloat16), device='cuda:0')
        bv = _pair_batch(2, 256, (90, 60), (40, 70), (80, 30), seed=13)
        for r in (2, 3):                        # the rows part right after BOS + the image tokens: nothing to share, each row would need its own features
            first = int((bv['attention_mask'][r] == 1).nonzero()[0])
            bv['input_ids'][r, first + 5] = (bv['input_ids'][r, first + 5] + 7) % 290 + 3
        assert tv._pack_plan(bv) is None
