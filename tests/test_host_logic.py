"""CPU: host-side logic of the native path (no compute calls): C-ABI loading/symbols, flat parameter
store <-> HF state dict, response-window index plan, LR schedule, config access."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests.util import ROOT, load_golden, state_dict_from_golden, tiny_llava_cfg, tiny_opt_cfg


def test_library_loads_and_exports_every_declared_symbol():
    from align_anything_amd import build, lib
    build.build()
    protos = {**lib.parse_header(lib.HEADER), **lib.parse_header(lib.HEADER_F32)}
    assert len(protos) >= 70
    dll = ctypes.CDLL(lib.LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), f'{name} declared in include/*.h but not exported'
    # and the other way round: every exported aa_* function is declared in a header (the fp32 twins of
    # elementwise.hip are produced by a second compilation, so the .so's dynamic symbol table is the ground truth)
    import subprocess
    nm = subprocess.run(['nm', '-D', '--defined-only', lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    defined = set(re.findall(r' T (aa_\w+)$', nm, flags=re.M))
    assert defined == set(protos), (defined ^ set(protos))
    twins = set(lib.parse_header(lib.HEADER_F32))
    assert {'aa_gemm_f32', 'aa_attn_fwd_f32', 'aa_attn_bwd_f32', 'aa_rmsnorm_fwd_f32', 'aa_layernorm_bwd_f32'} <= twins
    lib.LIB.load()
    assert lib.LIB._dll.aa_version() == 101      # 101: library contexts (aa_ctx_*)


def test_round6_switches_are_scoped_and_validated():
    """Two host-side rules of round 6 that need no device.  (1) ops.few_row_gemms: split-K for few-row GEMMs is on only inside the decorated RL-trainer
    methods (the preference trainers' pinned envelopes stay on the one-launch kernel), nests, and is restored when the call raises.  (2) aa_decode_set_rules
    hands back the previous mask and refuses masks it does not know."""
    from align_anything_amd import lib, ops
    from align_anything_amd.trainers import dpo, grpo, ppo, ppo_ti2t
    assert ops.SPLITK is False and ops._splitk_chunks(832, 3584, 18944) == 0
    seen = []

    @ops.few_row_gemms
    def inner(fail=False):
        seen.append((ops.SPLITK, ops._splitk_chunks(832, 3584, 18944), ops._splitk_chunks(4096, 3584, 18944), ops._splitk_chunks(832, 37888, 3584)))
        if fail:
            raise ValueError('x')

    @ops.few_row_gemms
    def outer():
        inner()
        seen.append(ops.SPLITK)

    outer()
    with pytest.raises(ValueError):
        inner(fail=True)
    assert ops.SPLITK is False
    allowed = ops.SPLITK_ALLOWED
    assert seen[0][0] is allowed and seen[1] is allowed
    if allowed:
        assert 2 <= seen[0][1] <= 8 and seen[0][2] == 0 and seen[0][3] == 0          # few rows x few tiles only: not M > 1024, not a launch that fills the chip
    for cls, names in ((ppo.PPOTrainer, ('actor_step', 'rollout', 'rl_step', 'ptx_step')), (ppo_ti2t.PPOTrainerTI2T, ('actor_step', 'rollout', 'rl_step')),
                       (grpo.GRPOTrainer, ('generate_completions', 'train_step', 'actor_step'))):
        for n in names:
            assert hasattr(getattr(cls, n), '__wrapped__'), (cls.__name__, n)
    assert not hasattr(dpo.DPOTrainer.train_step, '__wrapped__')
    old = ops.decode_set_rules(0)
    try:
        assert ops.decode_set_rules(3) == 0 and ops.decode_set_rules(1) == 3
        with pytest.raises(lib.AAHipError):
            ops.decode_set_rules(64)
        assert ops.decode_set_rules(1) == 1                                           # a refused mask changes nothing
    finally:
        ops.decode_set_rules(old)


def test_missing_library_fails_loudly(monkeypatch):
    from align_anything_amd import lib
    l = lib._Lib()
    monkeypatch.setattr(lib, 'LIB_PATH', '/nonexistent/libaa_hip.so')
    with pytest.raises(lib.AAHipError):
        l.load()


def test_ops_refuse_cpu_tensors():
    from align_anything_amd import ops
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(64, 64, dtype=torch.bfloat16))


def test_param_store_roundtrips_hf_state_dict_llava():
    from align_anything_amd.modeling import build_model
    z = load_golden('llava_tiny_dpo.npz')
    sd = state_dict_from_golden(z, 'w.', torch.bfloat16)
    m = build_model(tiny_llava_cfg(), 'cpu', trainable=True)
    missing = m.load_state_dict(sd)
    assert missing == []
    out = m.state_dict()
    assert set(out) == set(sd)
    for k in sd:
        assert out[k].shape == sd[k].shape, k
        assert torch.equal(out[k], sd[k]), k
    # fused blocks really are contiguous views of the HF tensors
    st = m.store
    q = st.view('model.language_model.layers.0.self_attn.q_proj.weight')
    blk = st.p['model.language_model.layers.0.self_attn.qkv_fused']
    assert q.data_ptr() == blk.data_ptr() and blk.shape == (3 * 128, 128)
    # optimizer grouping follows utils/tools.py:241-270
    assert st.specs['model.language_model.norm.weight']['group'] == 'vec'
    assert st.specs['model.multi_modal_projector.linear_1.bias']['group'] == 'vec'
    assert st.specs['lm_head.weight']['group'] == 'mat'
    assert st.specs['model.language_model.embed_tokens.weight']['group'] == 'emb'
    assert st.specs['model.vision_tower.post_layernorm.weight']['group'] == 'frozen'
    m.init_training()
    assert st.grad_view('model.language_model.layers.1.mlp.up_proj.weight').shape == (256, 128)
    assert st.master['mat'].dtype == torch.float32 and st.gflat['mat'].dtype == torch.bfloat16


def test_param_store_roundtrips_hf_state_dict_opt_with_tied_head():
    from align_anything_amd.modeling import build_model
    z = load_golden('opt_tiny_dpo.npz')
    sd = state_dict_from_golden(z, 'w.', torch.bfloat16)
    m = build_model(tiny_opt_cfg(), 'cpu')
    m.load_state_dict(sd)
    out = m.state_dict()
    assert set(out) == set(sd)
    for k in sd:
        assert torch.equal(out[k], sd[k]), k


def test_window_plan_matches_reference_slicing():
    """positions/labels of logits[idx][-R:][:-1] x strip_pad(ids)[-R:][1:] (dpo.py:131-139), CPU part."""
    from oracle import rl_math as orl
    from align_anything_amd.trainers import common
    z = load_golden('llava_tiny_dpo.npz')
    ids = torch.from_numpy(z['input_ids'])
    N, T = ids.shape
    lens = [int(x) for x in z['response_lens']]
    import align_anything_amd.ops as ops
    orig = ops.window_labels
    ops.window_labels = lambda *a, **k: None  # device kernel; checked in the gpu suite
    try:
        w = common.build_window(ids, lens, int(z['pad_token_id']))
    finally:
        ops.window_labels = orig
    off = 0
    for n, R in enumerate(lens):
        pos, _ = orl.response_window(ids[n], int(z['pad_token_id']), R, T)
        assert w['row_idx'][off:off + R - 1].tolist() == (pos + n * T).tolist()
        off += R - 1
    assert w['rows'] == sum(lens) - N and w['rows_pad'] % 64 == 0
    inv = w['inv_map']
    assert int((inv >= 0).sum()) == w['rows']
    assert inv[w['row_idx'][:w['rows']]].tolist() == list(range(w['rows']))
    flat = torch.arange(1, w['rows_pad'] + 1, dtype=torch.float32)
    padded = common.flat_to_padded(flat, w)
    assert padded.shape == (N, max(lens) - 1)
    for n, R in enumerate(lens):
        assert (padded[n, R - 1:] == 0).all() and (padded[n, :R - 1] != 0).all()
    with pytest.raises(ValueError):
        common.build_window(ids, [1, 2], 301)


def test_cosine_schedule_matches_hf_get_scheduler():
    from transformers import get_scheduler
    from align_anything_amd.engine import cosine_with_warmup
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1e-6)
    sch = get_scheduler('cosine', opt, num_warmup_steps=3, num_training_steps=40)
    for step in range(40):
        assert abs(opt.param_groups[0]['lr'] - cosine_with_warmup(step, 1e-6, 3, 40)) < 1e-15
        opt.step(); sch.step()


def test_cfg_get_reads_namedtuple_dict_and_missing():
    from collections import namedtuple
    from align_anything_amd.trainers.common import cfg_get
    Tc = namedtuple('Tc', ['scale_coeff'])
    C = namedtuple('C', ['train_cfgs'])
    c = C(Tc(0.2))
    assert cfg_get(c, 'train_cfgs.scale_coeff') == 0.2
    assert cfg_get(c, 'train_cfgs.missing', 7) == 7
    assert cfg_get({'a': {'b': 3}}, 'a.b') == 3 and cfg_get({'a': None}, 'a.b', 1) == 1


def test_dp_gradient_buckets_partition_the_flat_buffer():
    """NativeEngine all-reduces one contiguous slice of the flat bf16 gradient buffer per decoder layer plus the
    gaps (projector, lm_head): together they must cover every element exactly once."""
    from align_anything_amd.engine import NativeEngine
    from align_anything_amd.modeling import build_model
    for cfg in (tiny_llava_cfg(), tiny_opt_cfg()):
        m = build_model(cfg, 'cpu', trainable=True)
        eng = NativeEngine(m, lr=1e-6)
        n = m.store.gflat['mat'].numel()
        slices = sorted(eng._layer_slices.values())
        assert len(slices) == len(m.stack.layers)
        hits = torch.zeros(n, dtype=torch.int32)
        pos = 0
        for lo, hi in slices:
            assert lo >= pos, 'layer buckets overlap'
            if lo > pos:
                hits[pos:lo] += 1      # gap bucket issued after backward
            hits[lo:hi] += 1
            pos = hi
        if pos < n:
            hits[pos:] += 1
        assert int(hits.min()) == 1 and int(hits.max()) == 1
        # every matrix parameter of every layer lies inside its layer's bucket
        for L, (lo, hi) in zip(m.stack.layers, [eng._layer_slices[id(L)] for L in m.stack.layers]):
            for v in L.values():
                if hasattr(v, 'wname'):
                    s = m.store.specs[v.wname]
                    assert s['group'] != 'mat' or (lo <= s['offset'] and s['offset'] + s['numel'] <= hi)


def test_engine_lr_schedule_and_accumulation_bookkeeping():
    from align_anything_amd.engine import NativeEngine
    from align_anything_amd.modeling import build_model
    m = build_model(tiny_opt_cfg(), 'cpu', trainable=True)
    eng = NativeEngine(m, lr=1e-3, total_steps=10, warmup_steps=2, gradient_accumulation_steps=3)
    assert eng.optimizer.param_groups[0]['lr'] == 0.0          # cosine with warm-up starts at 0
    eng.micro_steps = 1
    eng.step()                                                  # not at a boundary: no update, no kernel call
    assert eng.global_steps == 0


def test_cached_input_pipeline_reproduces_the_reference_collator_batch():
    """align_anything_amd/data.py (tokenise once, collate from the cache, prefetch) vs the batch the reference's own
    PreferenceCollator produced on the same samples (tests/golden/collator.npz): integer work -> exact."""
    import numpy as np
    from oracle.synthetic import StubProcessor, preference_samples
    from align_anything_amd.data import CachedPreferenceCollator, DevicePrefetcher, TokenizedPreferenceCache
    from tests.util import load_golden
    z = load_golden('collator.npz')
    proc = StubProcessor()
    calls = {'n': 0}
    counting = lambda **kw: (calls.__setitem__('n', calls['n'] + 1), proc(**kw))[1]
    cache = TokenizedPreferenceCache(preference_samples(), counting)
    assert calls['n'] == 2 * len(cache)                                   # the processor ran once per conversation...
    for side in ('left', 'right'):
        batch = CachedPreferenceCollator(proc.pad_token_id, side)([cache[i] for i in range(len(cache))])
        assert np.array_equal(batch['input_ids'].numpy(), z[f'{side}_input_ids'])
        assert np.array_equal(batch['attention_mask'].numpy(), z[f'{side}_attention_mask'])
        assert np.array_equal(batch['pixel_values'].numpy(), z[f'{side}_pixel_values'])
        assert batch['meta_info']['response_lens'] == z[f'{side}_response_lens'].tolist()
    assert calls['n'] == 2 * len(cache)                                   # ...and never again while collating
    # prefetcher: same batches, same order, window plan attached (CPU path: no stream)
    coll = CachedPreferenceCollator(proc.pad_token_id, 'left')
    loader = [coll([cache[i], cache[i + 1]]) for i in range(0, 4, 2)]
    got = list(DevicePrefetcher(loader, 'cpu'))
    assert len(got) == 2 and all(torch.equal(g['input_ids'], b['input_ids']) for g, b in zip(got, loader))

    def boom():
        yield loader[0]
        raise RuntimeError('loader failed')
    with pytest.raises(RuntimeError, match='loader failed'):
        list(DevicePrefetcher(boom(), 'cpu'))


def test_span_and_tail_windows_match_the_reference_slicing():
    """trainers/common.py::build_span_window / build_tail_window are the index plans of
    `gather_log_probabilities(logits[:, :-1], ids[:, 1:])[:, start:]` (text_to_text/ppo.py:339-356) and of
    `logits[idx, :-1][-R:]` x `ids[idx, 1:][-R:]` (text_image_to_text/ppo.py:233-241): integer work, exact."""
    from align_anything_amd.trainers.common import build_span_window, build_tail_window, pad_rows
    g = torch.Generator().manual_seed(0)
    N, T, start = 3, 17, 5
    ids = torch.randint(0, 100, (N, T), generator=g)
    pos_code = torch.arange(N * T).view(N, T)                       # stands for "hidden state at (n, t)"
    w = build_span_window(ids, start)
    assert (w['N'], w['T'], w['W'], w['rows'], w['rows_pad']) == (N, T, T - 1 - start, N * (T - 1 - start), 64)
    rows = w['rows']
    assert torch.equal(w['row_idx'][:rows], pos_code[:, :-1][:, start:].reshape(-1))
    assert torch.equal(w['labels'][:rows], ids[:, 1:][:, start:].reshape(-1))
    inv = w['inv_map']
    assert inv.numel() == 64 and int((inv >= 0).sum()) == rows and torch.equal(inv[w['row_idx'][:rows]].long(), torch.arange(rows))
    with pytest.raises(ValueError):
        build_span_window(ids, T - 1)
    R = [4, 1, 16]
    t = build_tail_window(ids, R)
    assert t['rows'] == sum(R) and t['max_len'] == 16 and t['seq_off'].tolist() == [0, 4, 5, 21]
    for n in range(N):
        a, b = int(t['seq_off'][n]), int(t['seq_off'][n + 1])
        assert torch.equal(t['row_idx'][a:b], pos_code[n, :-1][-R[n]:])
        assert torch.equal(t['labels'][a:b], ids[n, 1:][-R[n]:])
        assert torch.equal(t['flat_to_padded'][a:b], n * 16 + torch.arange(R[n]))
    for bad in ([4, 1], [0, 1, 2], [4, 1, 17]):
        with pytest.raises(ValueError):
            build_tail_window(ids, bad)
    d = pad_rows(torch.ones(2, 3), 64)
    assert d.shape == (64,) and float(d.sum()) == 6.0 and float(d[6:].abs().sum()) == 0.0


def test_native_qwen2vl_rope_index_reproduces_hf_position_ids():
    """modeling.qwen2vl_rope_index (the product's host-side get_rope_index) against the position ids / rope deltas HF produced
    for the reference fixture: integer work, bit-exact; and the oracle's restatement agrees with both."""
    from align_anything_amd.modeling import qwen2vl_rope_index
    from oracle import models as om
    from tests.util import load_golden
    z = load_golden('qwen2vl_tiny_dpo.npz')
    ids, am = torch.from_numpy(z['input_ids']), torch.from_numpy(z['attention_mask'])
    grid = [[int(v) for v in r] for r in z['image_grid_thw']]
    from tests.util import tiny_qwen2vl_cfg
    cfg = tiny_qwen2vl_cfg()
    pos, deltas = qwen2vl_rope_index(ids, am, grid, cfg['image_token_id'], cfg['vision']['spatial_merge_size'])
    want = z['position_ids']
    valid = z['attention_mask'].astype(bool)
    assert pos.shape == want.shape
    assert (pos[:, valid] == want[:, valid]).all()                      # HF leaves pad positions at 1, the native table at 0: never read
    assert (deltas.reshape(-1) == z['rope_deltas'].reshape(-1)).all()
    o_pos, o_del = om.qwen2vl_rope_index(ids, am, grid, cfg['image_token_id'], cfg['vision']['spatial_merge_size'])
    assert (np.asarray(o_pos)[:, valid] == pos[:, valid]).all() and (np.asarray(o_del).reshape(-1) == deltas.reshape(-1)).all()
    with pytest.raises(ValueError):
        qwen2vl_rope_index(ids, am, [[1, 2, 2]] * len(grid), cfg['image_token_id'], cfg['vision']['spatial_merge_size'])


def test_from_hf_config_covers_every_native_backbone():
    """configs.from_hf_config reads what AnyModel.from_pretrained reads from config.json (models/model_registry.py:134-175)."""
    import transformers as tf
    from align_anything_amd import configs
    c = configs.from_hf_config(tf.OPTConfig(hidden_size=64, ffn_dim=128, num_hidden_layers=2, num_attention_heads=2, vocab_size=99, max_position_embeddings=77))
    assert (c['kind'], c['hidden_size'], c['ffn_dim'], c['num_layers'], c['num_heads'], c['vocab_size'], c['max_position_embeddings']) == ('opt', 64, 128, 2, 2, 99, 77)
    c = configs.from_hf_config(tf.LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2, vocab_size=321))
    assert (c['kind'], c['num_kv_heads'], c['head_dim'], c['attention_bias']) == ('llama', 2, 32, False)
    c = configs.from_hf_config(tf.Qwen2Config(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, vocab_size=321))
    assert c['kind'] == 'llama' and c['attention_bias'] is True and c['head_dim'] == 64
    moe = tf.Qwen3MoeConfig(hidden_size=128, moe_intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, vocab_size=320,
                            num_experts=8, num_experts_per_tok=2, head_dim=64, norm_topk_prob=True)
    c = configs.from_hf_config(moe)
    assert (c['kind'], c['num_experts'], c['num_experts_per_tok'], c['moe_intermediate_size'], c['head_dim'], c['norm_topk_prob']) == ('qwen3moe', 8, 2, 64, 64, True)
    moe.mlp_only_layers = [0]
    with pytest.raises(ValueError):
        configs.from_hf_config(moe)
    with pytest.raises(ValueError):
        configs.from_hf_config(tf.GPT2Config())
    # the named 7B geometries carry the published sizes
    assert configs.qwen2_audio_7b()['audio']['d_model'] == 1280 and configs.qwen2_audio_7b()['text']['vocab_size'] == 156032
    assert configs.qwen2_vl_7b()['text']['mrope_section'] == [16, 24, 24] and configs.llava_1_5_7b()['text']['vocab_size'] == 32064


def test_window_plans_on_random_ragged_batches():
    """300 random batches (left padding of any length, response lengths from 1 to the full row, single-row batches): the native
    index plans vs the reference's own slicing expressions evaluated literally -- integer work, exact."""
    from oracle import rl_math as orl
    from align_anything_amd.trainers import common
    import align_anything_amd.ops as ops
    g = torch.Generator().manual_seed(2024)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    orig = ops.window_labels
    ops.window_labels = lambda *a, **k: None              # the device half of build_window is checked in the gpu suite
    try:
        for _ in range(300):
            N, T, pad = ri(1, 5), ri(2, 40), 1
            ids = torch.randint(2, 50, (N, T), generator=g)
            lens = []
            for n in range(N):
                lp = ri(0, T - 1)
                ids[n, :lp] = pad
                lens.append(ri(1, T - lp))
            w = common.build_window(ids, lens, pad)
            code = torch.arange(N * T).view(N, T)
            off = 0
            for n, R in enumerate(lens):
                want = code[n][-R:][:-1]                     # logits[idx][-R:][:-1] reads these hidden positions (dpo.py:131-139)
                assert torch.equal(w['row_idx'][off:off + R - 1], want)
                pos, lab = orl.response_window(ids[n], pad, R, T)
                assert (pos + n * T).tolist() == want.tolist() and lab.tolist() == ids[n][ids[n] != pad][-R:][1:].tolist()
                off += R - 1
            assert w['rows'] == off and w['seq_off'].tolist() == [0] + list(np.cumsum([r - 1 for r in lens]))
            if w['rows']:
                assert torch.equal(w['inv_map'][w['row_idx'][:off]].long(), torch.arange(off))
            t = common.build_tail_window(ids, [min(r, T - 1) for r in lens])
            o2 = 0
            for n, R in enumerate(min(r, T - 1) for r in lens):
                assert torch.equal(t['row_idx'][o2:o2 + R], code[n, :-1][-R:]) and torch.equal(t['labels'][o2:o2 + R], ids[n, 1:][-R:])
                o2 += R
            if T >= 3:
                st = ri(0, T - 2)
                s = common.build_span_window(ids, st)
                assert torch.equal(s['row_idx'][:s['rows']], code[:, :-1][:, st:].reshape(-1)) and torch.equal(s['labels'][:s['rows']], ids[:, 1:][:, st:].reshape(-1))
            lab = ids.clone(); lab[ids == pad] = -100; lab[:, :ri(0, T - 1)] = -100
            if int((lab[:, 1:] != -100).sum()):
                lw = common.build_label_window(lab)
                n_i, j_i = (lab[:, 1:] != -100).nonzero(as_tuple=True)
                assert torch.equal(lw['row_idx'][:lw['rows']], n_i * T + j_i) and torch.equal(lw['labels'][:lw['rows']], lab[:, 1:][n_i, j_i])
    finally:
        ops.window_labels = orig


def test_every_evidence_file_cited_in_the_docs_exists():
    """DESIGN.md / README.md / INTEGRATION.md / profiles/README.md cite measured evidence and sources by path: none may dangle."""
    missing = []
    for doc in ('DESIGN.md', 'README.md', 'INTEGRATION.md', os.path.join('profiles', 'README.md')):
        text = open(os.path.join(ROOT, doc)).read()
        cited = [m.group(1) for m in re.finditer(r'`((?:profiles|tools|tests|oracle|align_anything_amd|include)/[^`\s]+)`', text)]
        cited += ['profiles/' + m.group(1) for m in re.finditer(r'`(r01_[^`\s]+)`', text)]
        for path in cited:
            path = path.split('::')[0].rstrip('.,;:)')
            if any(c in path for c in '…*<>{'):          # abbreviated / pattern citations
                continue
            if not os.path.exists(os.path.join(ROOT, path)):
                missing.append((doc, path))
    assert not missing, missing


def test_supervised_and_prompt_only_collators_reproduce_the_reference_batches():
    """data.py::SupervisedCollator / PromptOnlyCollator vs the reference's own collators (datasets/text_to_text/supervised.py:139-162,
    prompt_only.py:154-175; tests/golden/collator.npz): integer work, exact -- incl. the reference's quirk that a pad id INSIDE the text
    is masked out by the supervised collator (mask = ids != pad) but not by the prompt-only one; and the prefetcher attaches the
    label-window plan of the supervised loss."""
    from align_anything_amd.data import DevicePrefetcher, PromptOnlyCollator, SupervisedCollator
    z = load_golden('collator.npz')
    lens = [int(n) for n in z['tok_lens']]
    offs = np.concatenate([[0], np.cumsum(lens)])
    rows = [torch.from_numpy(z['tok_flat'][offs[i]:offs[i + 1]]) for i in range(len(lens))]
    labs = [torch.from_numpy(z['lab_flat'][offs[i]:offs[i + 1]]) for i in range(len(lens))]
    sb = SupervisedCollator(1)([{'input_ids': r, 'labels': l} for r, l in zip(rows, labs)])
    assert sb['attention_mask'].dtype == torch.bool
    for k in ('input_ids', 'labels', 'attention_mask'):
        assert torch.equal(sb[k], torch.from_numpy(z['sft_' + k])), k
    assert not bool(sb['attention_mask'][2, 4]) and int(sb['input_ids'][2, 4]) == 1
    pb = PromptOnlyCollator(1)([{'input_ids': r} for r in rows])
    for k in ('input_ids', 'attention_mask'):
        assert torch.equal(pb[k], torch.from_numpy(z['prompt_' + k])), k
    T = pb['input_ids'].shape[1]
    assert bool(pb['attention_mask'][2, T - lens[2] + 4]) and int(pb['input_ids'][2, T - lens[2] + 4]) == 1
    from align_anything_amd.data import UnmatchedSupervisedCollator
    ub = UnmatchedSupervisedCollator(1)([{'input_ids': r, 'response_lens': n} for r, n in zip(rows, (4, 1, 9, 2))])        # KTO's KL batches
    assert torch.equal(ub['input_ids'], torch.from_numpy(z['unmatched_input_ids'])) and torch.equal(ub['attention_mask'], torch.from_numpy(z['unmatched_attention_mask']))
    assert ub['meta_info']['response_lens'] == z['unmatched_response_lens'].tolist() and ub['labels'] is None and bool(z['unmatched_labels_is_none'])
    got = list(DevicePrefetcher([sb], 'cpu'))
    assert len(got) == 1 and '_labels_host' not in got[0]
    w = got[0]['_window']
    assert w['rows'] == int((sb['labels'][:, 1:] != -100).sum()) and torch.equal(got[0]['labels'], sb['labels'])


def test_qwen3moe_loads_hub_style_per_expert_checkpoints():
    """Hub checkpoints carry `mlp.experts.<e>.{gate,up,down}_proj.weight`; transformers >= 5 merges them on load
    (hf:conversion_mapping.py "qwen2_moe": stack on dim 0, gate before up).  The native loader does the same merge."""
    from align_anything_amd.modeling import build_model
    from tests.util import tiny_qwen3moe_cfg
    z = load_golden('qwen3moe_tiny_dpo.npz')
    sd = state_dict_from_golden(z, 'w.', torch.bfloat16)
    cfg = tiny_qwen3moe_cfg()
    F = cfg['moe_intermediate_size']
    hub = {k: v for k, v in sd.items() if '.mlp.experts.' not in k}
    for i in range(cfg['num_layers']):
        p = f'model.layers.{i}.mlp.experts.'
        for e in range(cfg['num_experts']):
            hub[f'{p}{e}.gate_proj.weight'] = sd[p + 'gate_up_proj'][e, :F].clone()
            hub[f'{p}{e}.up_proj.weight'] = sd[p + 'gate_up_proj'][e, F:].clone()
            hub[f'{p}{e}.down_proj.weight'] = sd[p + 'down_proj'][e].clone()
    a, b = build_model(cfg, 'cpu', trainable=False), build_model(cfg, 'cpu', trainable=False)
    assert a.load_state_dict(sd) == [] and b.load_state_dict(hub) == []
    sa, sb = a.state_dict(), b.state_dict()
    assert set(sa) == set(sb) == set(sd)
    for k in sd:
        assert torch.equal(sa[k], sd[k]) and torch.equal(sb[k], sd[k]), k
    # the same hub-style tensors through the lazy checkpoint reader (two safetensors shards + index): the per-expert keys of one layer are merged only when
    # the store asks for that layer's fused block (checkpoint.Derived), nothing else is materialised
    import json
    import tempfile
    import safetensors.torch as st
    from align_anything_amd.checkpoint import Derived, LazyCheckpoint
    with tempfile.TemporaryDirectory() as d:
        names = sorted(hub)
        halves = {'model-00001-of-00002.safetensors': names[: len(names) // 2], 'model-00002-of-00002.safetensors': names[len(names) // 2:]}
        for fn, ks in halves.items():
            st.save_file({k: hub[k].contiguous() for k in ks}, os.path.join(d, fn), metadata={'format': 'pt'})
        with open(os.path.join(d, 'model.safetensors.index.json'), 'w') as f:
            json.dump({'metadata': {}, 'weight_map': {k: fn for fn, ks in halves.items() for k in ks}}, f)
        lazy = LazyCheckpoint(d, 'qwen3moe')
        c = build_model(cfg, 'cpu', trainable=False)
        fused = c.fuse_expert_keys(lazy)
        assert isinstance(fused, Derived) and 'model.layers.0.mlp.experts.gate_up_proj' in fused and 'model.layers.0.mlp.experts.0.gate_proj.weight' not in fused
        assert c.load_state_dict(lazy) == []
        sc = c.state_dict()
        assert all(torch.equal(sc[k], sd[k]) for k in sd)
    del hub['model.layers.1.mlp.experts.3.up_proj.weight']
    with pytest.raises(KeyError):
        build_model(cfg, 'cpu', trainable=False).load_state_dict(hub)


def test_native_checkpoints_round_trip_through_hf_from_pretrained(tmp_path):
    """SURVEY section 8b checkpoint contract: what the native engine saves (`slice_<tag>/pytorch_model.bin` under HF parameter names +
    the HF config) must load with `from_pretrained` and give back the same weights -- OPT (tied head), Llama (GQA), Qwen3-MoE (fused
    expert tensors), LLaVA (CLIP patch embedding stored K-padded), Qwen2-VL, Qwen2-Audio, and through the trainer's own `save()`."""
    import transformers as tf
    from align_anything_amd import configs
    from align_anything_amd.engine import NativeEngine
    from align_anything_amd.modeling import build_model
    torch.manual_seed(0)
    cases = {
        'opt': (tf.OPTForCausalLM, tf.OPTConfig(hidden_size=128, ffn_dim=256, num_hidden_layers=2, num_attention_heads=2, vocab_size=320,
                                                max_position_embeddings=128, word_embed_proj_dim=128, dropout=0.0, pad_token_id=1)),
        'llama': (tf.LlamaForCausalLM, tf.LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                                                      num_key_value_heads=1, vocab_size=320, max_position_embeddings=256)),
        'qwen3_moe': (tf.Qwen3MoeForCausalLM, tf.Qwen3MoeConfig(hidden_size=128, moe_intermediate_size=64, intermediate_size=64, num_hidden_layers=2,
                                                                num_attention_heads=2, num_key_value_heads=1, vocab_size=320, num_experts=8,
                                                                num_experts_per_tok=2, head_dim=64, max_position_embeddings=256)),
    }
    vc = tf.CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=28, patch_size=14)
    tc = tf.LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2, vocab_size=320,
                        rms_norm_eps=1e-5, max_position_embeddings=256)
    cases['llava'] = (tf.LlavaForConditionalGeneration, tf.LlavaConfig(vision_config=vc, text_config=tc, image_token_id=300, image_seq_length=4))
    cases['qwen2_vl'] = (tf.Qwen2VLForConditionalGeneration, tf.Qwen2VLConfig(      # 80-wide vision heads: stored zero-padded natively, HF-shaped on disk
        text_config=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, vocab_size=320,
                         max_position_embeddings=256, rms_norm_eps=1e-6,
                         rope_parameters={'rope_type': 'default', 'rope_theta': 10000.0, 'mrope_section': [8, 12, 12]}),
        vision_config=dict(depth=2, embed_dim=320, hidden_size=128, num_heads=4, mlp_ratio=2, patch_size=14, temporal_patch_size=2,
                           spatial_merge_size=2, in_channels=3),
        image_token_id=300, video_token_id=301, vision_start_token_id=302, vision_end_token_id=303, bos_token_id=1, eos_token_id=2))
    cases['qwen2_audio'] = (tf.Qwen2AudioForConditionalGeneration, tf.Qwen2AudioConfig(
        audio_config=dict(num_mel_bins=64, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256, d_model=128, max_source_positions=32),
        text_config=dict(model_type='qwen2', hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                         num_key_value_heads=1, vocab_size=320, max_position_embeddings=256, rms_norm_eps=1e-6),
        audio_token_id=300))
    for name, (cls, hc) in cases.items():
        hf = cls(hc).eval()
        sd = {k: v.detach().clone() for k, v in hf.state_dict().items()}
        m = build_model(configs.from_hf_config(hf.config), 'cpu', trainable=False)
        assert m.load_state_dict(sd) == [] and set(m.state_dict()) == set(sd), name
        d = str(tmp_path / name)
        NativeEngine(m, trainable=False).save_16bit_model(d, 'pytorch_model.bin')
        hf.config.save_pretrained(d)
        back = cls.from_pretrained(d, torch_dtype=torch.bfloat16).state_dict()
        for k, v in sd.items():
            assert torch.equal(back[k].float(), v.to(torch.bfloat16).float()), (name, k)
    # safetensors variant (tied lm_head left out of the file, re-tied by HF on load)
    cls, hc = cases['opt']
    hf = cls(hc).eval()
    m = build_model(configs.from_hf_config(hc), 'cpu', trainable=False)
    m.load_state_dict(hf.state_dict())
    d = str(tmp_path / 'opt_st')
    NativeEngine(m, trainable=False).save_16bit_model(d, 'model.safetensors')
    hc.save_pretrained(d)
    back = cls.from_pretrained(d, torch_dtype=torch.bfloat16).state_dict()
    assert all(torch.equal(back[k].float(), v.to(torch.bfloat16).float()) for k, v in hf.state_dict().items())
    # the trainer's save(): slice_<tag>/ with config.json next to the weights
    from align_anything_amd.trainers.sft import SupervisedTrainer
    cls, hc = cases['opt']
    hf = cls(hc).eval()
    tr = SupervisedTrainer({'train_cfgs': {}, 'model_cfgs': {'pad_token_id': 1}, 'logger_cfgs': {'output_dir': str(tmp_path / 'out')}}, None,
                           model_cfg=configs.from_hf_config(hc), policy_state=hf.state_dict(), device='cpu')
    tr.hf_config = hc
    d = tr.save(tag=7)
    assert d.endswith('slice_7') and os.path.exists(os.path.join(d, 'config.json')) and os.path.exists(os.path.join(d, 'pytorch_model.bin'))
    back = cls.from_pretrained(d, torch_dtype=torch.bfloat16).state_dict()
    assert all(torch.equal(back[k].float(), v.to(torch.bfloat16).float()) for k, v in hf.state_dict().items())


def test_every_ctypes_call_site_matches_its_prototype_arity():
    """Static guard (no GPU needed): every `call('aa_...', ...)` in the host code passes exactly as many arguments as the header
    declares for that entry point (names built as 'aa_x' + suffix are matched by their constant prefix, starred arguments skipped)."""
    import ast
    import glob
    from align_anything_amd.lib import HEADER, HEADER_F32, parse_header
    protos = {**parse_header(HEADER), **parse_header(HEADER_F32)}

    def const_prefix(node):
        if isinstance(node, ast.Constant) and isinstance(node.value, str):
            return node.value
        if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Add):
            return const_prefix(node.left)
        if isinstance(node, ast.IfExp):            # 'aa_gemm_f32' if sfx else 'aa_gemm_bf16'
            return const_prefix(node.body)
        return None

    checked, bad = 0, []
    files = glob.glob(os.path.join(ROOT, 'align_anything_amd', '**', '*.py'), recursive=True) + [os.path.join(ROOT, 'bench.py'), os.path.join(ROOT, '__graft_entry__.py')]
    for f in files:
        for node in ast.walk(ast.parse(open(f).read())):
            if not (isinstance(node, ast.Call) and getattr(node.func, 'id', getattr(node.func, 'attr', None)) == 'call' and node.args):
                continue
            name = const_prefix(node.args[0])
            if not name or not name.startswith('aa_') or any(isinstance(a, ast.Starred) for a in node.args):
                continue
            cands = [n for n in protos if n == name or n.startswith(name)]
            assert cands, (os.path.basename(f), node.lineno, name)
            want = {len(protos[n][1]) for n in cands}
            checked += 1
            if len(node.args) - 1 not in want:
                bad.append((os.path.basename(f), node.lineno, name, len(node.args) - 1, sorted(want)))
    assert not bad, bad
    assert checked >= 70


def test_preference_cache_forwards_tokenizer_arguments_of_the_text_collator():
    """The text-to-text PreferenceCollator tokenises with add_special_tokens=False (datasets/text_to_text/preference.py:186-193): the cache
    must be able to make the same call, or every cached row would carry an extra BOS."""
    from align_anything_amd.data import CachedPreferenceCollator, TokenizedPreferenceCache

    class Tok:
        pad_token_id = 0

        def __call__(self, text, return_tensors='pt', add_special_tokens=True, **kw):
            ids = [1] * bool(add_special_tokens) + [10 + (ord(c) % 7) for c in text]
            return {'input_ids': torch.tensor([ids])}

    samples = [{'better_conversation': 'abcd', 'worse_conversation': 'xy', 'better_response_lens': 2, 'worse_response_lens': 1}]
    with_bos = TokenizedPreferenceCache(samples, Tok(), has_images=False)
    plain = TokenizedPreferenceCache(samples, Tok(), has_images=False, processor_kwargs={'add_special_tokens': False})
    assert with_bos[0]['better_ids'].tolist()[0] == 1 and len(with_bos[0]['better_ids']) == 5
    assert plain[0]['better_ids'].tolist() == [10 + (ord(c) % 7) for c in 'abcd'] and len(plain[0]['worse_ids']) == 2
    b = CachedPreferenceCollator(0, 'right')([plain[0]])
    assert b['input_ids'].shape == (2, 4) and b['attention_mask'].tolist() == [[1, 1, 1, 1], [1, 1, 0, 0]] and b['meta_info']['response_lens'] == [2, 1]


def _check_ring_gemm_isa(src_name, kernel_re, mfma, mfma_per_trip, min_kernels):
    import re
    import subprocess
    import tempfile
    from align_anything_amd import build as b
    src = os.path.join(b.CSRC, src_name)
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, 'k.s')
        r = subprocess.run([b.HIPCC, *b.FLAGS, '--cuda-device-only', '-S', src, '-o', asm, '-Rpass-analysis=kernel-resource-usage'],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        text = open(asm).read()
    names = re.findall(r'Function Name: (\S*' + kernel_re + r'\S*)', r.stderr)
    scratch = [int(x) for x in re.findall(r'ScratchSize \[bytes/lane\]: (\d+)', r.stderr)]
    vspill = [int(x) for x in re.findall(r'VGPRs Spill: (\d+)', r.stderr)]
    agprs = [int(x) for x in re.findall(r'AGPRs: (\d+)', r.stderr)]
    assert len(names) >= min_kernels and len(names) == len(scratch) == len(vspill) == len(agprs)
    assert all(s == 0 for s in scratch) and all(s == 0 for s in vspill), list(zip(names, scratch, vspill))
    assert all(a == 256 for a in agprs), agprs
    assert 'scratch_' not in text
    # between the first and the last MFMA of a kernel (= the K loop: four ring steps per trip) every vmcnt wait must be one of OURS, i.e.
    # sit inside an inline-asm block
    kernels = re.findall(r'^(_ZN\S*' + kernel_re + r'\S*):[^\n]*\n(.*?)\n\.Lfunc_end', text, flags=re.S | re.M)
    assert len(kernels) >= min_kernels
    for name, body in kernels:
        lines = body.split('\n')
        idx = [i for i, ln in enumerate(lines) if mfma in ln]
        assert len(idx) == mfma_per_trip, (name, len(idx))
        in_asm, compiler_waits, ours = False, 0, 0
        for ln in lines[idx[0]:idx[-1]]:
            if 'ASMSTART' in ln:
                in_asm = True
            elif 'ASMEND' in ln:
                in_asm = False
            elif 's_waitcnt' in ln and 'vmcnt' in ln:
                if in_asm:
                    ours += 1
                else:
                    compiler_waits += 1
            elif not in_asm and ('v_mov_b32' in ln or 'v_accvgpr' in ln):
                compiler_waits += 1000      # a register copy in the loop = the compiler is shuttling fragments / accumulators
        assert compiler_waits == 0, (name, compiler_waits)
        assert ours == 3, (name, ours)          # steps 1..3 of the trip (step 0's begin sits above the first MFMA)


def test_gemm4_kernels_keep_everything_in_registers():
    """csrc/gemm4.hip counts its own LDS reads and LDS-DMA requests (inline asm), which is only sound while the compiler neither spills
    nor keeps accumulators in scratch: a spill of a fragment register could store it before its untracked load has landed, and any
    scratch access makes hipcc put `s_waitcnt vmcnt(0)` into the K loop (draining the DMA ring every step -- measured -15 %).
    Compile the file and require: no scratch, no spills, accumulators in the accumulator file, and a K loop whose only vmcnt waits are
    the counted ones of the begin-of-step statements."""
    _check_ring_gemm_isa('gemm4.hip', r'gemm4(?:nt)?_kernel', 'v_mfma_f32_16x16x32_bf16', 256, 10)


def _scc_hazards(asm_text: str):
    """Instructions that READ SCC (s_cselect*, s_cbranch_scc*, s_addc / s_subb, s_cmov) whose reaching SCC definition, inside one basic block, is a
    scalar ALU instruction INSIDE an inline-asm block -- i.e. the compiler kept a compare / carry alive across a statement that overwrites it."""
    writers = ('s_cmp', 's_add_', 's_addc', 's_sub_', 's_subb', 's_and_', 's_or_', 's_xor_', 's_lshl', 's_lshr', 's_ashr', 's_andn2', 's_orn2', 's_nand', 's_nor', 's_xnor',
               's_min', 's_max', 's_bitcmp', 's_abs', 's_not', 's_bfe', 's_bcnt', 's_ff', 's_flbit', 's_wqm', 's_quadmask', 's_absdiff')
    readers = ('s_cselect', 's_cbranch_scc', 's_addc_u32', 's_subb_u32', 's_cmov')
    hits = []
    for m in re.finditer(r'^(_Z\S+):[^\n]*\n(.*?)\n\.Lfunc_end', asm_text, flags=re.S | re.M):
        in_asm, last = False, None
        for i, line in enumerate(m.group(2).split('\n')):
            ls = line.strip()
            if 'ASMSTART' in ls:
                in_asm = True
            elif 'ASMEND' in ls:
                in_asm = False
            elif not ls or ls.startswith(';'):
                continue
            elif ls.startswith('.L'):
                last = None                                  # a label: the reaching definition is not known from this walk
            else:
                op = ls.split()[0]
                if op.startswith(readers) and last == 'asm':
                    hits.append((m.group(1)[:70], i, ls))
                if op.startswith(writers):
                    last = 'asm' if in_asm else 'compiler'
    return hits


def test_no_scalar_condition_code_is_kept_alive_across_inline_asm():
    """Round 6: the LDS-DMA statements of gemm4.hip / attn128.inc contain `s_add_u32 m0, ...`, which writes SCC.  Without the clobber in their asm
    declaration hipcc split 64-bit address adds around them (s_add_u32 lo; <asm>; s_addc_u32 hi) in five shipped gemm4 kernels -- the high word then takes the
    asm's carry: wrong whenever an operand's first stages cross a 4 GB boundary.  Every asm statement that writes SCC now declares it (AA_SCC, aa_common.h);
    this test walks the generated ISA of every source with hand-written asm and fails on the pattern, and checks that the scanner still finds it in the
    old build (-DAA_NO_DECLARE_SCC)."""
    import subprocess
    import tempfile
    from align_anything_amd import build as b
    with tempfile.TemporaryDirectory() as d:
        def isa(src, extra=()):
            out = os.path.join(d, src + ('.old' if extra else '') + '.s')
            r = subprocess.run([b.HIPCC, *b.FLAGS, *extra, '--cuda-device-only', '-S', os.path.join(b.CSRC, src), '-o', out], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            return open(out).read()
        for src in ('gemm4.hip', 'attention.hip', 'gemm.hip', 'decode.hip', 'lmhead.hip'):
            assert _scc_hazards(isa(src)) == [], src
        assert len(_scc_hazards(isa('gemm4.hip', ('-DAA_NO_DECLARE_SCC=1',)))) >= 1          # the scanner sees what it is there to see


def test_attention_kernels_keep_fragments_in_registers_and_the_prefetch_in_flight():
    """csrc/attention.hip issues its transpose reads and DMA pieces as inline asm (the compiler neither tracks their completion nor orders
    them against each other).  That is only sound -- and only fast -- while (1) nothing spills: a spilled fragment register would be stored
    before its untracked read has landed; (2) the tile loops contain no compiler-made `s_waitcnt vmcnt` except the one glued to the
    end-of-tile barrier: the round-1 build had a full drain of the NEXT tile's DMA in the middle of every tile."""
    import subprocess
    import tempfile
    from align_anything_amd import build as b
    src = os.path.join(b.CSRC, 'attention.hip')
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, 'attention.s')
        r = subprocess.run([b.HIPCC, *b.FLAGS, '--cuda-device-only', '-S', src, '-o', asm, '-Rpass-analysis=kernel-resource-usage'],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        text = open(asm).read()
    names = re.findall(r'Function Name: (\S*attn_(?:fwd|bwd_dq|bwd_dkv)_kernel\S*)', r.stderr)
    blocks = re.split(r'Function Name: ', r.stderr)[1:]
    checked = 0
    for blk in blocks:
        if not re.match(r'\S*attn_(?:fwd|bwd_dq|bwd_dkv)_kernel', blk):
            continue
        assert int(re.search(r'ScratchSize \[bytes/lane\]: (\d+)', blk).group(1)) == 0, blk[:200]
        assert int(re.search(r'VGPRs Spill: (\d+)', blk).group(1)) == 0, blk[:200]
        checked += 1
    assert checked == 6 and len(names) == 6                 # forward, dQ, dK/dV x head_dim 64 / 128
    assert 'scratch_' not in text
    kernels = re.findall(r'^(_Z\d+attn_(?:fwd|bwd_dq|bwd_dkv)_kernel\S*):[^\n]*\n(.*?)\n\.Lfunc_end', text, flags=re.S | re.M)
    assert len(kernels) == 6
    for name, body in kernels:
        lines = [ln for ln in body.split('\n') if ln.strip() and not ln.strip().startswith(';') or 'ASM' in ln]
        mf = [i for i, ln in enumerate(lines) if 'v_mfma_f32_16x16x32_bf16' in ln]
        assert mf, name
        in_asm, stray = False, []
        for i in range(mf[0], mf[-1]):
            ln = lines[i]
            if 'ASMSTART' in ln:
                in_asm = True
            elif 'ASMEND' in ln:
                in_asm = False
            elif 's_waitcnt' in ln and 'vmcnt' in ln and not in_asm:
                if not any('s_barrier' in x for x in lines[i + 1:i + 3]):
                    stray.append(ln.strip())
        assert not stray, (name, stray)
        assert sum('ds_read_b64_tr_b16' in ln for ln in lines) > 0 and not any('ds_bpermute' in ln for ln in lines), name


def test_attention_block_orders_are_bijections_and_the_transpose_addresses_match():
    """Integer restatements of two pieces of csrc/attention.hip whose errors would be silent on most shapes: (1) q_block_of / kv_block_of
    (block index -> (sequence, head, block); XCD-local when Hkv * N % 8 == 0) must visit every (n, h, block) exactly once, all blocks of
    one kv head on ONE XCD and back to back in its dispatch sequence; (2) tr_lane_base + the XOR with (db << 5) + the row immediate must
    address the same bytes as the swizzled tile layout the DMA writes (unit_swz), for both head sizes."""
    def q_block_of(b, N, H, Hkv, nqb, xcd_local):
        HkN, group = Hkv * N, H // Hkv
        per = nqb * group
        if xcd_local and HkN % 8 == 0:
            hkn, r = (b & 7) + 8 * ((b >> 3) // per), (b >> 3) % per
        else:
            hkn, r = b % HkN, b // HkN
        n, hk = hkn // Hkv, hkn % Hkv
        return n, hk * group + r % group, hk, nqb - 1 - r // group

    def kv_block_of(b, N, Hkv, nkvb):
        HkN = Hkv * N
        if HkN % 8 == 0:
            hkn, kvb = (b & 7) + 8 * ((b >> 3) // nkvb), (b >> 3) % nkvb
        else:
            hkn, kvb = b % HkN, b // HkN
        return hkn // Hkv, hkn % Hkv, kvb

    for N, H, Hkv, T in ((8, 32, 32, 2048), (2, 8, 2, 1000), (3, 20, 20, 750), (4, 16, 16, 577), (8, 28, 4, 2048), (1, 32, 8, 130)):
        nqb, nkvb = (T + 127) // 128, (T + 63) // 64
        for local in (True, False):
            seen = [q_block_of(b, N, H, Hkv, nqb, local) for b in range(nqb * H * N)]
            assert len(set((n, h, qb) for n, h, _, qb in seen)) == nqb * H * N and all(hk == h // (H // Hkv) for _, h, hk, _ in seen)
            if local and (Hkv * N) % 8 == 0:
                for b, (n, h, hk, qb) in enumerate(seen):            # one kv head = one XCD, a contiguous stretch of its sequence
                    assert (n * Hkv + hk) % 8 == b % 8
        seen = [kv_block_of(b, N, Hkv, nkvb) for b in range(nkvb * Hkv * N)]
        assert len(set(seen)) == nkvb * Hkv * N
        if (Hkv * N) % 8 == 0:
            assert all((n * Hkv + hk) % 8 == b % 8 for b, (n, hk, _) in enumerate(seen))
    for HD in (64, 128):
        ROWB, DB = HD * 2, HD // 16
        swz = (lambda row: row & 7) if HD == 128 else (lambda row: (row >> 1) & 3)
        for lane in range(64):
            l15, g = lane & 15, lane >> 4
            row = 4 * g + (l15 >> 2)
            base = row * ROWB + (swz(row) << 5) + (l15 & 3) * 8                      # tr_lane_base
            for db in range(DB):
                for rows16 in range(4):
                    got = (base ^ (db << 5)) + rows16 * 16 * ROWB                    # tr_stream: XOR, then the immediate
                    r = rows16 * 16 + row
                    want = r * ROWB + ((db ^ swz(r)) << 5) + (l15 & 3) * 8           # lds_tr(): row r, unit db ^ swizzle(r), slot
                    assert got == want, (HD, lane, db, rows16)


def _tiny_llava_hf():
    import transformers as tf
    vc = tf.CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=28, patch_size=14)
    tc = tf.LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2, vocab_size=320,
                        rms_norm_eps=1e-5, max_position_embeddings=256)
    torch.manual_seed(3)
    return tf.LlavaForConditionalGeneration(tf.LlavaConfig(vision_config=vc, text_config=tc, image_token_id=300, image_seq_length=4)).eval()


def test_checkpoint_reader_streams_sharded_hf_checkpoints_bit_exact(tmp_path):
    """VERDICT r3 missing #2 / next #6: `model_cfgs.model_name_or_path` -> native buffers.  An HF-sharded tiny LLaVA (save_pretrained with a
    small max_shard_size: model-0000x-of-0000y.safetensors + index), the same weights under the transformers-4.x hub key layout
    (`language_model.model...`, `vision_tower.vision_model...`), as sharded pytorch_model-*.bin and as single files all load bit-exactly
    through checkpoint.LazyCheckpoint, one tensor at a time.  Reference: models/pretrained_model.py:304-311 (from_pretrained)."""
    import json
    import safetensors.torch as st
    from align_anything_amd import configs
    from align_anything_amd.checkpoint import LazyCheckpoint, load_pretrained, normalize_key
    from align_anything_amd.modeling import build_model
    hf = _tiny_llava_hf()
    want = {k: v.detach().clone() for k, v in hf.state_dict().items()}
    d5 = str(tmp_path / 'v5_sharded')
    hf.save_pretrained(d5, max_shard_size='200KB')
    assert os.path.exists(os.path.join(d5, 'model.safetensors.index.json')) and len([f for f in os.listdir(d5) if f.endswith('.safetensors')]) > 3
    lazy = LazyCheckpoint(d5, 'llava')
    assert set(lazy) == set(want) and not lazy._open.get('never')      # nothing but the index was read to list the names
    m = build_model(configs.from_hf_config(hf.config), 'cpu', trainable=False, dtype=torch.float32)
    assert m.load_state_dict(lazy) == []
    got = m.state_dict()
    assert set(got) == set(want) and all(torch.equal(got[k], want[k]) for k in want)
    # ---- the transformers 4.x hub layout of the same weights, sharded by hand (two files + index)
    old = {}
    for k, v in want.items():
        k4 = k
        if k.startswith('model.language_model.'):
            k4 = 'language_model.model.' + k[len('model.language_model.'):]
        elif k == 'lm_head.weight':
            k4 = 'language_model.lm_head.weight'
        elif k.startswith('model.vision_tower.'):
            k4 = 'vision_tower.vision_model.' + k[len('model.vision_tower.'):]
        elif k.startswith('model.multi_modal_projector.'):
            k4 = k[len('model.'):]
        assert normalize_key('llava', k4) == k, (k4, k)
        old[k4] = v.contiguous()
    d4 = tmp_path / 'v4_sharded'
    d4.mkdir()
    names = sorted(old)
    halves = {'model-00001-of-00002.safetensors': names[: len(names) // 2], 'model-00002-of-00002.safetensors': names[len(names) // 2:]}
    for fn, ks in halves.items():
        st.save_file({k: old[k] for k in ks}, str(d4 / fn), metadata={'format': 'pt'})
    (d4 / 'model.safetensors.index.json').write_text(json.dumps({'metadata': {}, 'weight_map': {k: fn for fn, ks in halves.items() for k in ks}}))
    hf.config.save_pretrained(str(d4))
    m4, tok, proc, hc = load_pretrained(str(d4), 'cpu', trainable=False, dtype=torch.float32)
    assert tok is None and proc is None and hc.model_type == 'llava'
    got = m4.state_dict()
    assert all(torch.equal(got[k], want[k]) for k in want)
    # ---- sharded .bin (mmap) and bf16 compute dtype: the flat buffers hold RNE(bf16) of the checkpoint
    db = tmp_path / 'bin_sharded'
    db.mkdir()
    for fn, ks in (('pytorch_model-00001-of-00002.bin', names[: len(names) // 2]), ('pytorch_model-00002-of-00002.bin', names[len(names) // 2:])):
        torch.save({k: old[k] for k in ks}, str(db / fn))
    (db / 'pytorch_model.bin.index.json').write_text(json.dumps({'metadata': {}, 'weight_map': {k: ('pytorch_model-00001-of-00002.bin' if k in names[: len(names) // 2] else 'pytorch_model-00002-of-00002.bin') for k in names}}))
    hf.config.save_pretrained(str(db))
    mb, *_ = load_pretrained(str(db), 'cpu', trainable=False, dtype=torch.bfloat16)
    got = mb.state_dict()
    assert all(torch.equal(got[k].float(), want[k].to(torch.bfloat16).float()) for k in want)
    # a shard named by the index but absent fails loudly, and so does a directory without weights
    os.remove(str(db / 'pytorch_model-00002-of-00002.bin'))
    with pytest.raises(FileNotFoundError):
        LazyCheckpoint(str(db), 'llava')
    with pytest.raises(FileNotFoundError):
        LazyCheckpoint(str(tmp_path), 'llava')


def test_load_pretrained_adds_the_pad_row_like_the_reference(tmp_path):
    """models/pretrained_model.py:61-157 resize_tokenizer_embedding: a tokenizer without a pad token gains `<pad>`, input and output embeddings grow
    by one row = the mean of the old rows, config.pad_token_id follows.  Checked against HF's own resize + the reference's mean-initialisation."""
    import transformers as tf
    from tokenizers import Tokenizer, models, pre_tokenizers
    from align_anything_amd.checkpoint import load_pretrained
    vocab = {w: i for i, w in enumerate(['<s>', '</s>', '<unk>'] + [f'w{i}' for i in range(61)])}
    tk = Tokenizer(models.WordLevel(vocab, unk_token='<unk>'))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    fast = tf.PreTrainedTokenizerFast(tokenizer_object=tk, bos_token='<s>', eos_token='</s>', unk_token='<unk>')
    assert fast.pad_token is None
    torch.manual_seed(5)
    hf = tf.LlamaForCausalLM(tf.LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1,
                                            vocab_size=64, max_position_embeddings=128, tie_word_embeddings=False)).eval()
    d = str(tmp_path / 'm')
    hf.save_pretrained(d)
    fast.save_pretrained(d)
    m, tok, proc, hc = load_pretrained(d, 'cpu', trainable=False, dtype=torch.float32, model_max_length=77)
    assert tok.pad_token == '<pad>' and tok.pad_token_id == 64 and len(tok) == 65 and tok.padding_side == 'left' and tok.model_max_length == 77
    assert hc.pad_token_id == 64 and hc.vocab_size == 65 and m.cfg['vocab_size'] == 65 and proc is None
    sd = m.state_dict()
    for k in ('model.embed_tokens.weight', 'lm_head.weight'):
        old = hf.state_dict()[k]
        assert sd[k].shape == (65, 128) and torch.equal(sd[k][:64], old) and torch.equal(sd[k][64], old.mean(dim=0))
    assert torch.equal(sd['model.layers.0.mlp.down_proj.weight'], hf.state_dict()['model.layers.0.mlp.down_proj.weight'])
    # a checkpoint whose embedding is LARGER than its tokenizer (72 rows, 64 tokens): the reference's resize_token_embeddings(len(tokenizer)) SHRINKS it
    # to 65 rows and writes the mean of the first 64 into the new one -- checked against transformers' own resize + the reference's initialisation
    torch.manual_seed(6)
    big = tf.LlamaForCausalLM(tf.LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1,
                                             vocab_size=72, max_position_embeddings=128, tie_word_embeddings=False)).eval()
    d2 = str(tmp_path / 'big')
    big.save_pretrained(d2)
    fast2 = tf.PreTrainedTokenizerFast(tokenizer_object=tk, bos_token='<s>', eos_token='</s>', unk_token='<unk>')
    fast2.save_pretrained(d2)
    m2, tok2, _, hc2 = load_pretrained(d2, 'cpu', trainable=False, dtype=torch.float32)
    want = tf.LlamaForCausalLM.from_pretrained(d2, dtype=torch.float32)
    want.resize_token_embeddings(65)
    for emb in (want.get_input_embeddings(), want.get_output_embeddings()):
        emb.weight.data[-1:] = emb.weight.data[:-1].mean(dim=0, keepdim=True)
    assert len(tok2) == 65 and hc2.vocab_size == 65 and m2.cfg['vocab_size'] == 65
    sd2 = m2.state_dict()
    assert torch.allclose(sd2['model.embed_tokens.weight'], want.get_input_embeddings().weight.data, atol=1e-7)
    assert torch.allclose(sd2['lm_head.weight'], want.get_output_embeddings().weight.data, atol=1e-7)
    # on a GPU such a row count is refused at load time, with the way out, not at the first step
    from align_anything_amd.checkpoint import check_vocab_rows
    check_vocab_rows(128256)
    with pytest.raises(RuntimeError, match='multiples of 4'):
        check_vocab_rows(128257, d2)


def test_library_contexts_isolate_switches_and_plans_per_thread():
    """include/aa_hip.h aa_ctx_*: the state behind the set_* switches / plan records / communicator belongs to the calling thread's current
    context; the default context keeps the old process-global behaviour.  Host-only calls (no kernel is launched: aa_gemm_glu_bwd_plan only
    consults the records and the argument alignment)."""
    import ctypes
    import threading
    from align_anything_amd.lib import LIB, AAHipError, call
    LIB.load()

    def plan():
        p = ctypes.c_int(-9)
        call('aa_gemm_glu_bwd_plan', 0x1000, 0x2000, 0x3000, 0x4000, 512, 11008, 4096, 4096, 11008, 22016, 22016, ctypes.addressof(p))
        return p.value

    def world():
        r, w = ctypes.c_int(-1), ctypes.c_int(-1)
        call('aa_comm_world', ctypes.byref(r), ctypes.byref(w))
        return r.value, w.value

    cur = ctypes.c_void_p(1)
    call('aa_ctx_get_current', ctypes.byref(cur))
    assert cur.value is None and world() == (0, 1)
    base = plan()
    assert base in (1, 2)                                   # fusable shape: no record yet (2) unless AA_GLU_BWD forces the fused kernel (1)
    c1, c2 = ctypes.c_void_p(), ctypes.c_void_p()
    call('aa_ctx_create', ctypes.byref(c1))
    call('aa_ctx_create', ctypes.byref(c2))
    try:
        call('aa_gemm_glu_bwd_set_mode', 0)                 # default context: always the unfused pair
        assert plan() == 0
        call('aa_ctx_set_current', c1)
        assert plan() == base                               # a fresh context is untouched by the default context's switch
        call('aa_gemm_glu_bwd_set_mode', 1)
        assert plan() == 1
        call('aa_ctx_set_current', c2)
        assert plan() == base
        call('aa_ctx_get_current', ctypes.byref(cur))
        assert cur.value == c2.value
        seen = {}
        t = threading.Thread(target=lambda: seen.update(other=plan()))      # the current context is per THREAD: a new thread is on the default one
        t.start(); t.join()
        assert seen['other'] == 0
        call('aa_ctx_set_current', None)
        assert plan() == 0
        with pytest.raises(AAHipError):
            call('aa_ctx_destroy', None)
    finally:
        call('aa_ctx_set_current', None)
        call('aa_gemm_glu_bwd_set_mode', -1)
        call('aa_ctx_destroy', c1)
        call('aa_ctx_destroy', c2)
    assert plan() == base


def test_checkpoint_reader_key_layouts_of_the_other_backbones(tmp_path):
    """checkpoint.normalize_key for Qwen2-VL and Qwen2-Audio: the transformers-4.x hub layout (`visual...` / `model.layers...`, `language_model.model...` /
    `audio_tower...`) and the >= 5 layout both load into the native stores bit-exactly (hf:conversion_mapping.py "Qwen2VLForConditionalGeneration", "qwen2_audio")."""
    import safetensors.torch as st
    import transformers as tf
    from align_anything_amd import configs
    from align_anything_amd.checkpoint import LazyCheckpoint, normalize_key
    from align_anything_amd.modeling import build_model
    torch.manual_seed(1)
    qvl = tf.Qwen2VLForConditionalGeneration(tf.Qwen2VLConfig(
        text_config=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, vocab_size=320,
                         max_position_embeddings=256, rms_norm_eps=1e-6, rope_parameters={'rope_type': 'default', 'rope_theta': 10000.0, 'mrope_section': [8, 12, 12]}),
        vision_config=dict(depth=2, embed_dim=320, hidden_size=128, num_heads=4, mlp_ratio=2, patch_size=14, temporal_patch_size=2, spatial_merge_size=2, in_channels=3),
        image_token_id=300, video_token_id=301, vision_start_token_id=302, vision_end_token_id=303, bos_token_id=1, eos_token_id=2)).eval()
    qa = tf.Qwen2AudioForConditionalGeneration(tf.Qwen2AudioConfig(
        audio_config=dict(num_mel_bins=64, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256, d_model=128, max_source_positions=32),
        text_config=dict(model_type='qwen2', hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, vocab_size=320,
                         max_position_embeddings=256, rms_norm_eps=1e-6), audio_token_id=300)).eval()

    def old_qvl(k):      # what transformers 4.x wrote
        if k.startswith('model.visual.'):
            return k[len('model.'):]
        if k.startswith('model.language_model.'):
            return 'model.' + k[len('model.language_model.'):]
        return k

    def old_qa(k):
        if k.startswith('model.language_model.'):
            return 'language_model.model.' + k[len('model.language_model.'):]
        if k == 'lm_head.weight':
            return 'language_model.lm_head.weight'
        if k.startswith('model.audio_tower.') or k.startswith('model.multi_modal_projector.'):
            return k[len('model.'):]
        return k

    for kind, hf, old in (('qwen2vl', qvl, old_qvl), ('qwen2audio', qa, old_qa)):
        want = {k: v.detach().clone() for k, v in hf.state_dict().items()}
        d5 = str(tmp_path / (kind + '_v5'))
        hf.save_pretrained(d5, max_shard_size='300KB')
        d4 = tmp_path / (kind + '_v4')
        d4.mkdir()
        legacy = {old(k): v.contiguous() for k, v in want.items()}
        assert len(legacy) == len(want) and all(normalize_key(kind, old(k)) == k for k in want), kind
        assert any(old(k) != k for k in want)
        st.save_file(legacy, str(d4 / 'model.safetensors'), metadata={'format': 'pt'})
        for d in (d5, str(d4)):
            m = build_model(configs.from_hf_config(hf.config), 'cpu', trainable=False, dtype=torch.float32)
            lazy = LazyCheckpoint(d, kind)
            assert m.load_state_dict(lazy) == [], (kind, d)
            got = m.state_dict()
            assert set(got) == set(want) and all(torch.equal(got[k], want[k]) for k in want), (kind, d)


def test_checkpoint_reader_loads_what_save_pretrained_writes_for_every_backbone(tmp_path):
    """Whatever key layout the installed transformers WRITES (it reverses some of its load-time renames on save -- Qwen2-VL and Qwen2-Audio come out in their 4.x
    layouts, Qwen3-MoE's experts fused or per expert depending on the version) must come back bit-exactly through LazyCheckpoint + the model's loader."""
    import transformers as tf
    from align_anything_amd import configs
    from align_anything_amd.checkpoint import LazyCheckpoint
    from align_anything_amd.modeling import build_model
    torch.manual_seed(2)
    cases = {
        'opt': tf.OPTForCausalLM(tf.OPTConfig(hidden_size=128, ffn_dim=256, num_hidden_layers=2, num_attention_heads=2, vocab_size=320, max_position_embeddings=128,
                                              word_embed_proj_dim=128, dropout=0.0, pad_token_id=1)),
        'llama': tf.LlamaForCausalLM(tf.LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, vocab_size=320,
                                                    max_position_embeddings=256)),
        'qwen3moe': tf.Qwen3MoeForCausalLM(tf.Qwen3MoeConfig(hidden_size=128, moe_intermediate_size=64, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                                                             num_key_value_heads=1, vocab_size=320, num_experts=8, num_experts_per_tok=2, head_dim=64, max_position_embeddings=256)),
        'llava': _tiny_llava_hf(),
    }
    for kind, hf in cases.items():
        hf = hf.eval()
        want = {k: v.detach().clone() for k, v in hf.state_dict().items()}
        d = str(tmp_path / kind)
        hf.save_pretrained(d, max_shard_size='150KB')
        m = build_model(configs.from_hf_config(hf.config), 'cpu', trainable=False, dtype=torch.float32)
        assert m.kind == kind
        assert m.load_state_dict(LazyCheckpoint(d, kind)) == [], kind
        got = m.state_dict()
        assert set(got) == set(want) and all(torch.equal(got[k], want[k]) for k in want), kind


def test_device_prefetcher_retires_the_producer_of_an_abandoned_iterator():
    """`next(iter(loader))`, a `break` or an exception in the step leave a prefetch iterator unfinished: its producer thread must stop (it would hold
    `depth` staged batches forever, and run the dataset -- a non-re-entrant fast tokenizer -- next to the producer of the following iterator)."""
    import threading
    import time
    from align_anything_amd.data import DevicePrefetcher
    active, peak, lock = [0], [0], threading.Lock()

    class Loader:
        def __len__(self):
            return 50

        def __iter__(self):
            for i in range(50):
                with lock:
                    active[0] += 1
                    peak[0] = max(peak[0], active[0])
                time.sleep(0.002)
                with lock:
                    active[0] -= 1
                yield {'input_ids': torch.full((2, 3), i)}

    pf = DevicePrefetcher(Loader(), 'cpu', depth=2)
    before = threading.active_count()
    first = next(iter(pf))
    assert int(first['input_ids'][0, 0]) == 0
    for i, b in enumerate(pf):
        if i == 3:
            break
    got = [int(b['input_ids'][0, 0]) for b in pf]
    assert got == list(range(50)) and peak[0] == 1            # never two producers inside the dataset at once
    with pytest.raises(ZeroDivisionError):
        for b in pf:
            1 / 0
    deadline = time.time() + 5
    while threading.active_count() > before and time.time() < deadline:
        time.sleep(0.01)
    assert threading.active_count() == before


def test_rope_scaling_matches_hf_and_unknown_types_raise():
    """configs.rope_scaling_of + modeling.rope_inv_freq against transformers' own rotary embedding: 'llama3' (meta-llama/Llama-3.1-8B-Instruct's
    published parameters: the reference's text-to-text default, scripts/llama/*.sh) and 'linear'; a type that is not built raises instead of
    being dropped (wrong logits at every position otherwise)."""
    from types import SimpleNamespace
    import transformers as tf
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding
    from align_anything_amd import configs
    from align_anything_amd.modeling import rope_inv_freq, rope_tables
    base = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=2, vocab_size=64, max_position_embeddings=131072)
    cases = {'llama3': {'rope_type': 'llama3', 'rope_theta': 500000.0, 'factor': 8.0, 'low_freq_factor': 1.0, 'high_freq_factor': 4.0, 'original_max_position_embeddings': 8192},
             'linear': {'rope_type': 'linear', 'rope_theta': 10000.0, 'factor': 4.0},
             'default': {'rope_type': 'default', 'rope_theta': 10000.0}}
    for name, rp in cases.items():
        hf_cfg = tf.LlamaConfig(**base, rope_parameters=dict(rp))
        cfg = configs.from_hf_config(hf_cfg)
        assert (cfg['rope_scaling'] or {}).get('type') == (None if name == 'default' else name)
        rot = LlamaRotaryEmbedding(hf_cfg)
        inv = rope_inv_freq(cfg['head_dim'], cfg['rope_theta'], cfg['rope_scaling'])
        assert torch.equal(inv, rot.inv_freq.float()), (name, float((inv - rot.inv_freq).abs().max()))
        pos = torch.tensor([[0, 1, 17, 4095, 8191, 8192, 60000]])
        cos, sin = rot(torch.zeros(1, 7, 8), pos)
        mine_c, mine_s = rope_tables(60001, cfg['head_dim'], cfg['rope_theta'], 'cpu', torch.float32, cfg['rope_scaling'])
        assert torch.allclose(mine_c[pos[0]], cos[0, :, :cfg['head_dim'] // 2], atol=1e-6) and torch.allclose(mine_s[pos[0]], sin[0, :, :cfg['head_dim'] // 2], atol=1e-6), name
    assert not torch.equal(rope_inv_freq(128, 500000.0, configs.rope_scaling_of(tf.LlamaConfig(**base, rope_parameters=dict(cases['llama3'])))), rope_inv_freq(128, 500000.0))
    # transformers-4.x spelling: `rope_scaling` with the key 'type'
    old = SimpleNamespace(rope_parameters=None, rope_scaling={'type': 'linear', 'factor': 2.0}, max_position_embeddings=4096)
    assert configs.rope_scaling_of(old) == {'type': 'linear', 'factor': 2.0}
    with pytest.raises(ValueError, match='yarn'):
        configs.rope_scaling_of(SimpleNamespace(rope_parameters={'rope_type': 'yarn', 'factor': 4.0}, max_position_embeddings=4096))
    with pytest.raises(ValueError, match='post-layer-norm'):
        configs.from_hf_config(tf.OPTConfig(hidden_size=64, ffn_dim=128, num_hidden_layers=1, num_attention_heads=2, vocab_size=50, do_layer_norm_before=False))
    with pytest.raises(ValueError, match='mlp_bias'):
        configs.from_hf_config(tf.LlamaConfig(**base, mlp_bias=True))


def test_llama_family_checkpoint_with_tied_embeddings_loads_and_saves(tmp_path):
    """config.tie_word_embeddings (Llama-3.2-1B / 3B, Qwen2.5-0.5B ... 3B): the checkpoint has no lm_head tensor; the native model keeps ONE
    parameter for both roles (as NativeOPT does), `state_dict()` shows it under both names, and a save -> HF load round trip keeps the tie."""
    import transformers as tf
    from align_anything_amd.checkpoint import load_pretrained
    torch.manual_seed(0)
    hf_cfg = tf.Qwen2Config(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, vocab_size=320,
                            max_position_embeddings=256, tie_word_embeddings=True)
    hf = tf.Qwen2ForCausalLM(hf_cfg).eval()
    d = str(tmp_path / 'tied')
    hf.save_pretrained(d)
    m, _, _, _ = load_pretrained(d, 'cpu', trainable=True, with_tokenizer=False)
    assert m.tied and m.cfg['attention_bias'] and 'lm_head.weight' not in m.store.specs and m.head.lm_w == m.embed
    sd = m.state_dict()
    assert torch.equal(sd['lm_head.weight'], sd['model.embed_tokens.weight'])
    assert torch.equal(sd['model.embed_tokens.weight'].float(), hf.state_dict()['model.embed_tokens.weight'].to(torch.bfloat16).float())
    m.init_training()
    assert m.store.g[m.embed].dtype == torch.float32          # the shared gradient buffer: head dW accumulates into it, the embedding scatter-adds
    m.load_state_dict(sd)                                      # its own state dict (with the alias) loads back
    # save -> transformers loads it back with the tie intact (both file formats of save_16bit_model)
    from align_anything_amd.engine import NativeEngine
    eng = NativeEngine(m, trainable=False)
    for fn in ('model.safetensors', 'pytorch_model.bin'):
        out = str(tmp_path / ('saved_' + fn.split('.')[0]))
        eng.save_16bit_model(out, save_filename=fn)
        hf_cfg.save_pretrained(out)
        back = tf.Qwen2ForCausalLM.from_pretrained(out, dtype=torch.float32)
        assert back.lm_head.weight.data_ptr() == back.model.embed_tokens.weight.data_ptr()
        assert torch.equal(back.model.embed_tokens.weight.float(), sd['model.embed_tokens.weight'].float())
        assert torch.equal(back.model.layers[1].mlp.down_proj.weight.float(), sd['model.layers.1.mlp.down_proj.weight'].float())
    untied = tf.Qwen2Config(**{**hf_cfg.to_dict(), 'tie_word_embeddings': False})
    d2 = str(tmp_path / 'untied')
    tf.Qwen2ForCausalLM(untied).save_pretrained(d2)
    m2, _, _, _ = load_pretrained(d2, 'cpu', trainable=False, with_tokenizer=False)
    assert not m2.tied and 'lm_head.weight' in m2.store.specs


def test_device_prefetcher_nested_iteration_fails_instead_of_hanging():
    from align_anything_amd.data import DevicePrefetcher
    pf = DevicePrefetcher([{'input_ids': torch.full((1, 2), i)} for i in range(6)], 'cpu', depth=1)
    outer = iter(pf)
    next(outer)
    assert [int(b['input_ids'][0, 0]) for b in pf] == list(range(6))       # a second iteration retires the first one's producer ...
    with pytest.raises(RuntimeError, match='retired'):
        for _ in range(6):
            next(outer)                                                      # ... whose consumer is told so (after the batches already staged)


def test_trainers_expose_the_reference_s_phase_methods():
    """SURVEY.md section 8(b), trainer surface: the reference's trainers are built from init_check / init_models / init_datasets / init_engines /
    init_logger (text_to_text/dpo.py:59-77, rm.py:57-67, ppo.py:62-91, grpo.py:69-75) and its modality subclasses override exactly those; the
    native trainers keep the names so such a subclass ports over."""
    from align_anything_amd.trainers.dpo import DPOTrainer
    from align_anything_amd.trainers.grpo import GRPOTrainer
    from align_anything_amd.trainers.ppo import PPOTrainer
    from align_anything_amd.trainers.ppo_ti2t import PPOTrainerTI2T
    from align_anything_amd.trainers.rm import RMTrainer
    from align_anything_amd.trainers.sft import SupervisedTrainer
    phases = ['init_check', 'init_models', 'init_datasets', 'init_engines', 'init_logger']
    for cls in (DPOTrainer, SupervisedTrainer, RMTrainer, PPOTrainer, PPOTrainerTI2T, GRPOTrainer):
        assert all(callable(getattr(cls, n, None)) for n in phases + ['train', 'eval', 'save']), cls.__name__
    for cls in (DPOTrainer, SupervisedTrainer, RMTrainer):
        assert all(callable(getattr(cls, n, None)) for n in ('loss', 'train_step')), cls.__name__
    for cls in (PPOTrainer, PPOTrainerTI2T):
        assert all(callable(getattr(cls, n, None)) for n in ('rollout', 'rl_step', 'ptx_step', 'actor_step', 'set_train', 'split_ptx_micro_batches', 'actor_loss_fn',
                                                             'critic_loss_fn', 'add_kl_divergence_regularization', 'get_advantages_and_returns')), cls.__name__
    assert all(callable(getattr(GRPOTrainer, n, None)) for n in ('generate_completions', 'compute_rewards', 'train_step', 'set_train'))
    # the phases run in the reference's order, and an overridden phase is the one that runs
    order = []

    class Probe(RMTrainer):
        def init_check(self):
            order.append('check')
            super().init_check()

        def init_models(self, state=None):
            order.append('models')
            self.module, self._from_path = None, False

        def init_datasets(self):
            order.append('datasets')

        def init_engines(self):
            order.append('engines')

        def init_logger(self):
            order.append('logger')

    Probe({'train_cfgs': {}, 'model_cfgs': {'model_name_or_path': 'x'}}, None, device='cpu')
    assert order == ['check', 'models', 'datasets', 'engines', 'logger']
    pb = {'input_ids': torch.arange(6).reshape(3, 2), 'labels': torch.arange(6).reshape(3, 2), 'meta_info': {'k': 1}}
    mb = PPOTrainer.split_ptx_micro_batches(pb)
    assert len(mb) == 3 and mb[1]['input_ids'].tolist() == [[2, 3]] and mb[2]['meta_info'] == {'k': 1}


def test_checkpoint_tensors_without_a_native_home_raise(tmp_path):
    """A checkpoint tensor the native model cannot place (here: an MLP bias a llama config did not announce, saved by hand) must stop the load;
    benign extras (the rotary buffer of old checkpoints, the lm_head of a language model loaded as a score model) pass."""
    import safetensors.torch as st
    import transformers as tf
    from align_anything_amd.checkpoint import load_pretrained
    torch.manual_seed(0)
    hf = tf.LlamaForCausalLM(tf.LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=2, vocab_size=64,
                                            max_position_embeddings=64))
    d = str(tmp_path / 'm')
    hf.save_pretrained(d)
    f = d + '/model.safetensors'
    sd = st.load_file(f)
    sd['model.layers.0.self_attn.rotary_emb.inv_freq'] = torch.ones(32)
    st.save_file(sd, f, metadata={'format': 'pt'})
    load_pretrained(d, 'cpu', trainable=False, with_tokenizer=False)                      # the rotary buffer is benign
    m = load_pretrained(d, 'cpu', trainable=False, head='score', with_tokenizer=False)[0]   # ... and so is lm_head for a score model
    assert 'score_head.weight' in m.store.specs
    sd['model.layers.0.mlp.gate_proj.bias'] = torch.zeros(256)
    st.save_file(sd, f, metadata={'format': 'pt'})
    with pytest.raises(RuntimeError, match='does not implement'):
        load_pretrained(d, 'cpu', trainable=False, with_tokenizer=False)


def test_published_config_layouts_of_the_baseline_models_parse(tmp_path):
    """config.json as the hub holds it for BASELINE configs[1] / [2] (written by transformers 4.36 / 4.41: `image_token_index`, a FLAT qwen2_vl text
    config with `rope_scaling: {type: mrope}`, `in_chans`): AutoConfig + configs.from_hf_config give the 7B geometries the benchmarks run."""
    import json
    import transformers as tf
    from align_anything_amd import configs
    llava = {"architectures": ["LlavaForConditionalGeneration"], "ignore_index": -100, "image_token_index": 32000, "model_type": "llava", "pad_token_id": 32001,
             "projector_hidden_act": "gelu", "text_config": {"_name_or_path": "lmsys/vicuna-7b-v1.5", "architectures": ["LlamaForCausalLM"], "max_position_embeddings": 4096,
                                                             "model_type": "llama", "rms_norm_eps": 1e-05, "torch_dtype": "float16", "vocab_size": 32064},
             "tie_word_embeddings": False, "torch_dtype": "float16", "transformers_version": "4.36.0.dev0",
             "vision_config": {"hidden_size": 1024, "image_size": 336, "intermediate_size": 4096, "model_type": "clip_vision_model", "num_attention_heads": 16,
                               "num_hidden_layers": 24, "patch_size": 14, "projection_dim": 768, "vocab_size": 32000},
             "vision_feature_layer": -2, "vision_feature_select_strategy": "default", "vocab_size": 32064}
    qwen = {"architectures": ["Qwen2VLForConditionalGeneration"], "attention_dropout": 0.0, "bos_token_id": 151643, "eos_token_id": 151645, "vision_start_token_id": 151652,
            "vision_end_token_id": 151653, "vision_token_id": 151654, "image_token_id": 151655, "video_token_id": 151656, "hidden_act": "silu", "hidden_size": 3584,
            "initializer_range": 0.02, "intermediate_size": 18944, "max_position_embeddings": 32768, "max_window_layers": 28, "model_type": "qwen2_vl",
            "num_attention_heads": 28, "num_hidden_layers": 28, "num_key_value_heads": 4, "rms_norm_eps": 1e-06, "rope_theta": 1000000.0, "sliding_window": 32768,
            "tie_word_embeddings": False, "torch_dtype": "bfloat16", "transformers_version": "4.41.2", "use_cache": True, "use_sliding_window": False,
            "vision_config": {"depth": 32, "embed_dim": 1280, "mlp_ratio": 4, "num_heads": 16, "in_chans": 3, "hidden_size": 3584, "patch_size": 14, "spatial_merge_size": 2,
                              "spatial_patch_size": 14, "temporal_patch_size": 2},
            "rope_scaling": {"type": "mrope", "mrope_section": [16, 24, 24]}, "vocab_size": 152064}
    out = {}
    for name, cfg in (('llava', llava), ('qwen2vl', qwen)):
        d = tmp_path / name
        d.mkdir()
        (d / 'config.json').write_text(json.dumps(cfg))
        out[name] = configs.from_hf_config(tf.AutoConfig.from_pretrained(str(d)))
    c = out['llava']
    assert c == configs.llava_1_5_7b() or (c['image_token_id'], c['vision_feature_layer'], c['pad_token_id']) == (32000, -2, 32001)
    t, v = c['text'], c['vision']
    assert (t['hidden_size'], t['intermediate_size'], t['num_layers'], t['num_heads'], t['num_kv_heads'], t['head_dim'], t['vocab_size'], t['rms_eps'], t['rope_theta']) == \
        (4096, 11008, 32, 32, 32, 128, 32064, 1e-5, 10000.0) and t['rope_scaling'] is None
    assert (v['hidden_size'], v['intermediate_size'], v['num_layers'], v['num_heads'], v['image_size'], v['patch_size']) == (1024, 4096, 24, 16, 336, 14)
    c = out['qwen2vl']
    t, v = c['text'], c['vision']
    assert (t['hidden_size'], t['intermediate_size'], t['num_layers'], t['num_heads'], t['num_kv_heads'], t['head_dim'], t['vocab_size'], t['rope_theta'], t['mrope_section']) == \
        (3584, 18944, 28, 28, 4, 128, 152064, 1000000.0, [16, 24, 24]) and t['attention_bias'] and c['image_token_id'] == 151655
    assert (v['embed_dim'], v['depth'], v['num_heads'], v['hidden_size'], v['patch_size'], v['temporal_patch_size'], v['spatial_merge_size'], v['in_channels']) == \
        (1280, 32, 16, 3584, 14, 2, 2, 3)


def test_lora_and_quantised_training_are_refused_not_ignored():
    """`lora_cfgs.use_lora` / `bnb_cfgs.use_bnb` change what the reference trains (base/supervised_trainer.py:53-58); the native trainers implement
    neither and say so in init_check instead of silently fine-tuning the full bf16 weights."""
    from align_anything_amd.trainers.dpo import DPOTrainer
    from align_anything_amd.trainers.grpo import GRPOTrainer
    from align_anything_amd.trainers.ppo import PPOTrainer
    from align_anything_amd.trainers.rm import RMTrainer
    for cls in (DPOTrainer, RMTrainer, PPOTrainer, GRPOTrainer):
        for bad in ({'lora_cfgs': {'use_lora': True}}, {'bnb_cfgs': {'use_bnb': True}}, {'train_cfgs': {'fp16': True}}):
            with pytest.raises(NotImplementedError):
                cls(dict({'train_cfgs': {}, 'model_cfgs': {}}, **bad), None, model_cfg=tiny_opt_cfg(), device='cpu')


def test_lr_schedules_equal_transformers_get_scheduler():
    """base/supervised_trainer.py:251-257 builds `get_scheduler(name=train_cfgs.lr_scheduler_type, ...)`: the native engine's closed forms for cosine, linear,
    constant and constant_with_warmup against transformers' own schedulers, step by step."""
    import transformers as tf
    from types import SimpleNamespace
    from align_anything_amd.engine import NativeEngine
    for name in ('cosine', 'linear', 'constant', 'constant_with_warmup'):
        for warm, total in ((0, 10), (3, 17)):
            p = torch.nn.Parameter(torch.zeros(1))
            opt = torch.optim.SGD([p], lr=2e-3)
            sch = tf.get_scheduler(name=name, optimizer=opt, num_warmup_steps=warm, num_training_steps=total)
            eng = SimpleNamespace(sched=name, base_lr=2e-3, warmup_steps=warm, total_steps=total)
            for step in range(total + 1):
                want = opt.param_groups[0]['lr']
                assert abs(NativeEngine._lr_at(eng, step) - want) < 1e-12, (name, warm, total, step, NativeEngine._lr_at(eng, step), want)
                opt.step()
                sch.step()
    with pytest.raises(ValueError, match='not supported'):
        NativeEngine._lr_at(SimpleNamespace(sched='polynomial', base_lr=1.0, warmup_steps=0, total_steps=5), 1)


def test_power_sampler_summary_and_header(tmp_path):
    """tools/power_sampler.py (bench.py's sidecar): `summarise` keeps only the samples inside the timed region and derives the clock-scaled peak; without any
    SMI source (this container) the sidecar writes a header naming what failed and exits non-zero instead of hanging -- bench.py then reports nulls."""
    import json
    import subprocess
    import sys
    from tools.power_sampler import summarise
    f = tmp_path / 's.jsonl'
    rows = [{'header': {'source': 'amdsmi', 'errors': {}, 'hz': 20.0}}] + \
           [{'t': 100.0 + 0.05 * i, 'power_w': 1000.0 + i, 'sclk_mhz': 1800.0 + 10 * (i % 3)} for i in range(40)]
    f.write_text('\n'.join(json.dumps(r) for r in rows) + '\nnot json\n')
    s = summarise(str(f), 100.5, 101.0, peak_tflops=2500.0)
    inside = [r for r in rows[1:] if 100.5 <= r['t'] <= 101.0]
    assert s['power_samples'] == len(inside) == s['sclk_samples'] and s['power_source'] == 'amdsmi'
    assert abs(s['power_w_mean'] - sum(r['power_w'] for r in inside) / len(inside)) < 1e-9 and s['power_w_max'] == max(r['power_w'] for r in inside)
    assert abs(s['peak_at_sclk'] - 2500.0 * s['sclk_mhz_mean'] / 2400.0) < 1e-9 and s['sclk_mhz_min'] == 1800.0
    empty = summarise(str(f), 500.0, 501.0)
    assert empty['power_w_mean'] is None and empty['sclk_mhz_mean'] is None and 'peak_at_sclk' not in empty
    assert summarise(str(tmp_path / 'missing.jsonl'), 0, 1)['power_w_mean'] is None
    out = tmp_path / 'live.jsonl'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'power_sampler.py'), '--out', str(out), '--seconds', '0.2'], timeout=60)
    hdr = json.loads(out.read_text().split('\n')[0])['header']
    if hdr['source'] is None:          # no GPU here: every source reports why
        assert r.returncode != 0 and set(hdr['errors']) == {'amdsmi', 'sysfs', 'smi'}
