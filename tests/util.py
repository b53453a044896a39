"""Shared helpers for the test-suite (fixtures <-> torch)."""
import ast
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def bits_to_bf16(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.astype(np.int16)).view(torch.bfloat16)


def load_golden(name: str):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def state_dict_from_golden(z, prefix='w.', dtype=torch.float32):
    return {k[len(prefix):]: bits_to_bf16(z[k]).to(dtype) for k in z.files if k.startswith(prefix)}


def tiny_llava_cfg():
    from align_anything_amd import configs
    text = configs.llama_cfg(128, 256, 2, 2, 2, 320, rms_eps=1e-5, max_position_embeddings=256)
    vision = configs.clip_vision_cfg(128, 256, 3, 2, 28, 14)
    return configs.llava_cfg(text, vision, image_token_id=300, pad_token_id=301)


def tiny_opt_cfg():
    from align_anything_amd import configs
    return configs.opt_cfg(128, 256, 2, 2, 320, 128)


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.double(); b = b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def tiny_qwen2vl_cfg():
    from align_anything_amd import configs
    text = configs.llama_cfg(128, 256, 2, 2, 1, 320, rms_eps=1e-6, rope_theta=10000.0, max_position_embeddings=256, attention_bias=True)
    vision = configs.qwen2vl_vision_cfg(320, 2, 4, 2, 128)
    return configs.qwen2vl_cfg(text, vision, image_token_id=300, mrope_section=[8, 12, 12], pad_token_id=304)


def tiny_qwen2audio_cfg():
    from align_anything_amd import configs
    text = configs.llama_cfg(128, 256, 2, 2, 1, 320, rms_eps=1e-6, rope_theta=10000.0, max_position_embeddings=256, attention_bias=True)
    audio = configs.qwen2audio_tower_cfg(128, 2, 2, 256, num_mel_bins=64, max_source_positions=32)
    return configs.qwen2audio_cfg(text, audio, audio_token_id=300, pad_token_id=304)


def tiny_qwen3moe_cfg():
    from align_anything_amd import configs
    return configs.qwen3moe_cfg(128, 64, 2, 2, 1, 320, 8, 2, head_dim=64, rope_theta=10000.0, max_position_embeddings=256)


def tiny_llava_checkpoint(path: str, seed: int = 0):
    """A LLaVA checkpoint directory as `save_pretrained` writes it, built offline: HF LlavaForConditionalGeneration (CLIP tower 3 x 128 on 28 x 28 /
    14 images -> 4 image tokens, Llama 2 x 128) with random weights, plus a REAL `LlavaProcessor` (PIL CLIP image processor, word-level fast
    tokenizer with `<image>` / `<pad>` tokens, a chat template that renders {'type': 'image'} parts as `<image>`).  Returns (hf model, processor)."""
    import transformers as tf
    from tokenizers import Tokenizer, models, pre_tokenizers
    vocab = {w: i for i, w in enumerate(['<s>', '</s>', '<unk>', '<pad>', '<image>'] + [f'w{i}' for i in range(315)])}
    tk = Tokenizer(models.WordLevel(vocab, unk_token='<unk>'))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    fast = tf.PreTrainedTokenizerFast(tokenizer_object=tk, bos_token='<s>', eos_token='</s>', unk_token='<unk>', pad_token='<pad>')
    fast.add_special_tokens({'additional_special_tokens': ['<image>']})
    from transformers.models.clip.image_processing_pil_clip import CLIPImageProcessorPil
    ip = CLIPImageProcessorPil(size={'shortest_edge': 28}, crop_size={'height': 28, 'width': 28})
    template = ("{% for m in messages %}{{ m['role'] }} : {% for c in m['content'] %}{% if c['type']=='image' %}<image> {% else %}{{ c['text'] }}{% endif %}"
                "{% endfor %} {% if m['role']=='assistant' %}</s> {% endif %}{% endfor %}{% if add_generation_prompt %}assistant :{% endif %}")
    processor = tf.LlavaProcessor(image_processor=ip, tokenizer=fast, patch_size=14, vision_feature_select_strategy='default',
                                  num_additional_image_tokens=1, chat_template=template)
    vc = tf.CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=28, patch_size=14, projection_dim=64)
    tc = tf.LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2, vocab_size=320,
                        rms_norm_eps=1e-5, max_position_embeddings=256)
    cfg = tf.LlavaConfig(vision_config=vc, text_config=tc, image_token_id=4, image_seq_length=4, pad_token_id=3)
    torch.manual_seed(seed)
    hf = tf.LlavaForConditionalGeneration(cfg).eval()
    hf.save_pretrained(path)
    processor.save_pretrained(path)
    return hf, processor


def ti2t_parquet_dataset(path: str, n: int = 12, seed: int = 0) -> str:
    """A local dataset directory `datasets.load_dataset(path)` reads, in the raw layout the reference's `AA_TI2T` template formats
    (configs/format_dataset.py:465-560): question, image (an Image feature -> PIL), response_1, response_2, overall_response."""
    import os
    import datasets
    from PIL import Image
    rng = np.random.RandomState(seed)
    words = lambda k: ' '.join(f'w{int(i)}' for i in rng.randint(0, 300, k))
    ds = datasets.Dataset.from_dict({
        'question': [words(rng.randint(2, 7)) for _ in range(n)],
        'image': [Image.fromarray((rng.rand(30 + i, 40, 3) * 255).astype('uint8')) for i in range(n)],
        'response_1': [words(rng.randint(1, 9)) for _ in range(n)], 'response_2': [words(rng.randint(1, 9)) for _ in range(n)],
        'overall_response': [1 + i % 2 for i in range(n)]}).cast_column('image', datasets.Image())
    os.makedirs(path, exist_ok=True)
    ds.to_parquet(os.path.join(path, 'train.parquet'))
    return path


# ---------------------------------------------------------------------------------------------------------------------------------------
# The end-to-end drop-in fixture (tests/golden/dropin_e2e.npz, oracle/gen_golden.py::gen_dropin_e2e)
DROPIN_SPECIALS = ['<s>', '</s>', '<unk>', '<pad>']
DROPIN_TEMPLATE = "{% for m in messages %}{{ m['role'] }} : {{ m['content'] }} </s> {% endfor %}"


def dropin_tokenizer(words):
    """Word-level fast tokenizer: the four specials + `words` (ids 4 ..), whitespace pre-tokenisation, left padding as the reference's DPO trainer
    loads it (text_to_text/dpo.py:94).  The fixture generator passes the most frequent words of the reference's asset file; the GPU test, which
    only needs the ids the generator stored, passes anonymous names of the same count."""
    import transformers as tf
    from tokenizers import Tokenizer, models, pre_tokenizers
    vocab = {w: i for i, w in enumerate(DROPIN_SPECIALS + list(words))}
    tk = Tokenizer(models.WordLevel(vocab, unk_token='<unk>'))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    fast = tf.PreTrainedTokenizerFast(tokenizer_object=tk, bos_token='<s>', eos_token='</s>', unk_token='<unk>', pad_token='<pad>', padding_side='left',
                                      model_max_length=512)
    fast.chat_template = DROPIN_TEMPLATE
    return fast


def dropin_hf_config(vocab_size):
    import transformers as tf
    return tf.OPTConfig(hidden_size=128, ffn_dim=256, num_hidden_layers=2, num_attention_heads=2, vocab_size=int(vocab_size), max_position_embeddings=600,
                        word_embed_proj_dim=128, dropout=0.0, attention_dropout=0.0, pad_token_id=3, bos_token_id=0, eos_token_id=1)


def dropin_checkpoint(path: str, z) -> None:
    """The HF checkpoint directory the drop-in test hands to `model_cfgs.model_name_or_path`: config.json + model.safetensors (the fixture's weights,
    bf16-representable fp32) + the tokenizer files (anonymous vocabulary of the fixture's size)."""
    import transformers as tf
    cfg = dropin_hf_config(int(z['vocab_size']))
    hf = tf.OPTForCausalLM(cfg)
    sd = state_dict_from_golden(z, 'w.', torch.float32)
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not unexpected and set(missing) <= {'lm_head.weight'}, (missing, unexpected)
    hf.save_pretrained(path)
    dropin_tokenizer([f'w{i}' for i in range(int(z['vocab_size']) - len(DROPIN_SPECIALS))]).save_pretrained(path)


class DropinPreferenceDataset:
    """Stand-in for `align_anything.datasets.text_to_text.PreferenceDataset` where the reference package is absent (the GPU box): same constructor
    signature, `get_collator()`, `__len__` / `__getitem__` -- but the samples are the PRE-TOKENISED output of the reference's own dataset + template +
    tokenizer on its own asset file (the fixture's `b_ids` / `w_ids`), so the collator pads ids where the reference's tokenises text.  tests/
    test_dropin_cpu.py checks, in the build container, that its batches equal the real plugin's batch for batch."""

    def __init__(self, path, template, tokenizer, processor=None, name=None, size=None, split=None, data_files=None, optional_args=[]):
        z = np.load(path)
        self.tokenizer = tokenizer
        cut = lambda flat, off: [flat[off[i]:off[i + 1]].astype(np.int64) for i in range(len(off) - 1)]
        self.b, self.w = cut(z['b_ids'], z['b_off']), cut(z['w_ids'], z['w_off'])
        self.bl, self.wl = z['b_resp_len'], z['w_resp_len']

    def __len__(self):
        return len(self.b)

    def __getitem__(self, i):
        return {'better_ids': self.b[i], 'worse_ids': self.w[i], 'better_response_lens': int(self.bl[i]), 'worse_response_lens': int(self.wl[i])}

    def get_collator(self):
        pad, left = int(self.tokenizer.pad_token_id), self.tokenizer.padding_side == 'left'

        def collate(samples):
            rows = [s['better_ids'] for s in samples] + [s['worse_ids'] for s in samples]        # chosen rows, then rejected rows (preference.py:186-188)
            L = max(len(r) for r in rows)
            ids = torch.full((len(rows), L), pad, dtype=torch.long)
            am = torch.zeros((len(rows), L), dtype=torch.long)
            for i, r in enumerate(rows):
                sl = slice(L - len(r), L) if left else slice(0, len(r))
                ids[i, sl] = torch.from_numpy(r)
                am[i, sl] = 1
            return {'input_ids': ids, 'attention_mask': am,
                    'meta_info': {'response_lens': [s['better_response_lens'] for s in samples] + [s['worse_response_lens'] for s in samples]}}
        return collate


def install_dropin_plugins(monkeypatch) -> None:
    """Register the stand-in plugin package under the names `common.get_dataloaders` imports (align_anything.datasets.text_to_text,
    align_anything.configs.template) for the duration of a test."""
    import sys
    import types
    pk = {n: types.ModuleType(n) for n in ('align_anything', 'align_anything.datasets', 'align_anything.datasets.text_to_text', 'align_anything.configs',
                                           'align_anything.configs.template')}
    for m in pk.values():
        m.__path__ = []
    pk['align_anything.datasets.text_to_text'].PreferenceDataset = DropinPreferenceDataset
    pk['align_anything.configs.template'].ChatTemplate = lambda formatter, name, custom=None: types.SimpleNamespace(formatter=formatter, name=name)
    for n, m in pk.items():
        monkeypatch.setitem(sys.modules, n, m)


# ---- the text+image sibling (tests/golden/dropin_e2e_ti2t.npz, oracle/gen_golden.py::gen_dropin_e2e_ti2t)
def dropin_ti2t_checkpoint(path: str, z) -> None:
    """The LLaVA checkpoint directory of the text+image drop-in test: tests/util.tiny_llava_checkpoint's files (config, a REAL LlavaProcessor with its
    tokenizer / image processor / chat template) with the fixture's weights."""
    import transformers as tf
    hf, _ = tiny_llava_checkpoint(path, seed=5)
    sd = state_dict_from_golden(z, 'w.', torch.float32)
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not unexpected and not [m for m in missing if 'lm_head' not in m], (missing, unexpected)
    hf.save_pretrained(path)


class DropinTI2TPreferenceDataset(DropinPreferenceDataset):
    """Stand-in for `align_anything.datasets.text_image_to_text.PreferenceDataset` (see DropinPreferenceDataset): the reference's pre-processed samples -- token
    ids AND the LlavaProcessor's pixel values; the collator stacks the images twice (chosen rows, then rejected rows), as preference.py:219-222 does."""

    def __init__(self, path, template, tokenizer, processor=None, name=None, size=None, split=None, data_files=None, optional_args=[]):
        super().__init__(path, template, tokenizer, processor, name, size, split, data_files, optional_args)
        self.pix = np.load(path)['pixel_values']

    def __getitem__(self, i):
        d = super().__getitem__(i)
        d['pixel_values'] = self.pix[i]
        return d

    def get_collator(self):
        base = super().get_collator()

        def collate(samples):
            out = base(samples)
            pv = torch.from_numpy(np.stack([s['pixel_values'] for s in samples]))
            out['pixel_values'] = torch.cat([pv, pv], 0)
            return out
        return collate


def install_dropin_ti2t_plugins(monkeypatch) -> None:
    import sys
    import types
    install_dropin_plugins(monkeypatch)
    m = types.ModuleType('align_anything.datasets.text_image_to_text')
    m.__path__ = []
    m.PreferenceDataset = DropinTI2TPreferenceDataset
    monkeypatch.setitem(sys.modules, 'align_anything.datasets.text_image_to_text', m)


def dropin_rm_checkpoint(path: str, z) -> None:
    """The reward-model checkpoint directory of the RM drop-in test: OPT config.json + model.safetensors holding the backbone AND `score_head.weight` (the layout the
    reference's AccustomedOPTRewardModel saves, models/opt.py:31-97) + the anonymous word-level tokenizer."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    cfg = dropin_hf_config(int(z['vocab_size']))
    cfg.architectures = ['OPTForCausalLM']
    cfg.save_pretrained(path)
    sd = state_dict_from_golden(z, 'w.', torch.float32)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, 'model.safetensors'), metadata={'format': 'pt'})
    dropin_tokenizer([f'w{i}' for i in range(int(z['vocab_size']) - len(DROPIN_SPECIALS))]).save_pretrained(path)


class DropinSupervisedDataset:
    """Stand-in for `align_anything.datasets.text_to_text.SupervisedDataset` (see DropinPreferenceDataset): the reference's PRE-TOKENISED samples (ids + labels with
    the prompt at -100); the collator right-pads ids with the pad token and labels with -100 and derives the mask from the ids, as supervised.py:134-157 does."""

    def __init__(self, path, template, tokenizer, processor=None, name=None, size=None, split=None, data_files=None, optional_args=[]):
        z = np.load(path)
        off = z['off']
        self.ids = [z['ids'][off[i]:off[i + 1]].astype(np.int64) for i in range(len(off) - 1)]
        self.labels = [z['labels'][off[i]:off[i + 1]].astype(np.int64) for i in range(len(off) - 1)]
        self.pad = int(tokenizer.pad_token_id)

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, i):
        return {'input_ids': torch.from_numpy(self.ids[i]), 'labels': torch.from_numpy(self.labels[i])}

    def get_collator(self):
        pad = self.pad

        def collate(samples):
            L = max(len(s['input_ids']) for s in samples)
            ids = torch.full((len(samples), L), pad, dtype=torch.long)
            lab = torch.full((len(samples), L), -100, dtype=torch.long)
            for i, s in enumerate(samples):
                ids[i, :len(s['input_ids'])] = s['input_ids']
                lab[i, :len(s['labels'])] = s['labels']
            return {'input_ids': ids, 'labels': lab, 'attention_mask': ids.ne(pad)}
        return collate


def install_dropin_sft_plugins(monkeypatch) -> None:
    import sys
    install_dropin_plugins(monkeypatch)
    sys.modules['align_anything.datasets.text_to_text'].SupervisedDataset = DropinSupervisedDataset


class DropinPromptOnlyDataset:
    """Stand-in for `align_anything.datasets.text_to_text.PromptOnlyDataset`: the PROMPT of every pre-tokenised preference sample (its tokens before the chosen
    response); the collator left-pads, as prompt_only.py's PromptOnlyCollator does."""

    def __init__(self, path, template, tokenizer, processor=None, name=None, size=None, split=None, data_files=None, optional_args=[]):
        z = np.load(path)
        off = z['b_off']
        self.rows = [z['b_ids'][off[i]:off[i + 1] - int(z['b_resp_len'][i])].astype(np.int64) for i in range(len(off) - 1)]
        if size:
            self.rows = self.rows[:int(size)]
        self.pad = int(tokenizer.pad_token_id)

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i):
        return {'input_ids': torch.from_numpy(self.rows[i])}

    def get_collator(self):
        pad = self.pad

        def collate(samples):
            L = max(len(s['input_ids']) for s in samples)
            ids = torch.full((len(samples), L), pad, dtype=torch.long)
            am = torch.zeros((len(samples), L), dtype=torch.bool)
            for i, s in enumerate(samples):
                ids[i, L - len(s['input_ids']):] = s['input_ids']
                am[i, L - len(s['input_ids']):] = True
            return {'input_ids': ids, 'attention_mask': am}
        return collate


def install_dropin_rl_plugins(monkeypatch) -> None:
    import sys
    install_dropin_plugins(monkeypatch)
    sys.modules['align_anything.datasets.text_to_text'].PromptOnlyDataset = DropinPromptOnlyDataset
    sys.modules['align_anything.datasets.text_to_text'].SupervisedDataset = DropinSupervisedDataset
