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
    assert lib.LIB._dll.aa_version() == 100


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
