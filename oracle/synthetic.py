"""TEST INFRASTRUCTURE ONLY.  Synthetic workload definitions shared by oracle/gen_golden.py (which runs the reference on
them, in the build container) and the GPU tests (which run the native path on them).  Pure HF + torch CPU RNG: this
module never touches /root/reference, so it can be imported on the GPU box."""
from __future__ import annotations

import torch


def opt125m_config1(num_pairs=64, T=256, R=128):
    """BASELINE.md section 2, config 1 (== BASELINE.json configs[0]): OPT-125m geometry, dropout 0, seed 42,
    reference = policy + N(0, 1e-3), `num_pairs` synthetic pairs of T tokens (no padding) with R response tokens.
    Returns (policy, reference, batches).  Pure HF + torch CPU RNG, so tests/test_f32_gpu.py rebuilds the identical
    weights on the GPU box (the 500 MB of weights are not committed; a per-tensor checksum is)."""
    from transformers import OPTConfig, OPTForCausalLM
    oc = OPTConfig(dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, layerdrop=0.0)
    torch.manual_seed(42)
    policy = OPTForCausalLM(oc).eval()
    torch.manual_seed(42)
    refm = OPTForCausalLM(oc).eval()
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for p in refm.parameters():
            p.add_(1e-3 * torch.randn(p.shape, generator=g))
    gb = torch.Generator().manual_seed(1234)
    batches = []
    for _ in range(num_pairs):
        ids = torch.randint(3, oc.vocab_size, (2, T), generator=gb)
        ids[1, :T - R] = ids[0, :T - R]                  # chosen / rejected share the prompt
        batches.append({'input_ids': ids, 'attention_mask': torch.ones_like(ids), 'meta_info': {'response_lens': [R, R]}})
    return oc, policy, refm, batches


class StubProcessor:
    """A deterministic stand-in for an HF multimodal processor (no tokenizer files are available offline): words ->
    ids by a fixed hash, `<image>` -> `n_image_tokens` image-token ids, BOS in front; `padding=True` pads the batch to
    its longest row on `padding_side`; images (ints used as seeds) -> pixel tensors.  Used both by the reference's
    unmodified PreferenceCollator (oracle/gen_golden.py::gen_collator) and by the native cached pipeline."""

    def __init__(self, vocab=300, image_token_id=300, pad_token_id=301, n_image_tokens=4, image_size=28):
        self.vocab, self.image_token_id, self.pad_token_id = vocab, image_token_id, pad_token_id
        self.n_image_tokens, self.image_size = n_image_tokens, image_size

    def _ids(self, text):
        out = [1]
        for w in text.split():
            if w == '<image>':
                out += [self.image_token_id] * self.n_image_tokens
            else:
                h = 0
                for ch in w:
                    h = (h * 131 + ord(ch)) % 1000003
                out.append(3 + h % (self.vocab - 3))
        return out

    def _pixels(self, image):
        g = torch.Generator().manual_seed(int(image))
        return torch.randn(3, self.image_size, self.image_size, generator=g)

    def __call__(self, text=None, images=None, return_tensors='pt', padding=False, padding_side='right',
                 return_attention_mask=True, **kw):
        texts = [text] if isinstance(text, str) else list(text)
        rows = [self._ids(t) for t in texts]
        T = max(len(r) for r in rows)
        ids = torch.full((len(rows), T), self.pad_token_id, dtype=torch.int64)
        mask = torch.zeros((len(rows), T), dtype=torch.int64)
        for r, row in enumerate(rows):
            if padding_side == 'left':
                ids[r, T - len(row):] = torch.tensor(row); mask[r, T - len(row):] = 1
            else:
                ids[r, :len(row)] = torch.tensor(row); mask[r, :len(row)] = 1
        out = {'input_ids': ids, 'attention_mask': mask}
        if images is not None:
            imgs = [images] if not isinstance(images, (list, tuple)) else list(images)
            out['pixel_values'] = torch.stack([self._pixels(i) for i in imgs])
        return out


def preference_samples(n=5, seed=3):
    """What PreferenceDataset.preprocess returns (datasets/text_image_to_text/preference.py:132-160), synthetic."""
    g = torch.Generator().manual_seed(seed)
    words = ['alpha', 'beta', 'gamma', 'delta', 'eps', 'zeta', 'eta', 'theta', 'iota', 'kappa', 'lam', 'mu']
    pick = lambda k: ' '.join(words[int(i)] for i in torch.randint(0, len(words), (k,), generator=g))
    out = []
    for i in range(n):
        prompt = 'USER: <image> ' + pick(int(torch.randint(2, 9, (1,), generator=g))) + ' ASSISTANT:'
        nb, nw = int(torch.randint(1, 12, (1,), generator=g)), int(torch.randint(1, 12, (1,), generator=g))
        out.append({'better_conversation': prompt + ' ' + pick(nb) + ' </s>', 'worse_conversation': prompt + ' ' + pick(nw) + ' </s>',
                    'image': 100 + i, 'better_response_lens': nb + 1, 'worse_response_lens': nw + 1})
    return out


def llava7b_width_config(num_layers=4):
    from transformers import CLIPVisionConfig, LlamaConfig, LlavaConfig
    vc = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336, patch_size=14,
                          projection_dim=768)
    tc = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=num_layers, num_attention_heads=32, num_key_value_heads=32,
                     vocab_size=32064, rms_norm_eps=1e-5, max_position_embeddings=4096)
    return LlavaConfig(vision_config=vc, text_config=tc, image_token_id=32000, image_seq_length=576, pad_token_id=32001)


def _width_tensor(name, shape, dim, seed, reference=False, trainable_matrix=True):
    """One tensor of a *_width fixture from its OWN generator, seeded by (seed, crc32 of the parameter name): matrices, embeddings and biases N(0, 0.02),
    norm weights 1 + N(0, 0.1), rounded to bf16-representable values; the reference model's copy = policy + N(0, 2e-3) (drawn AFTER the policy's numbers from
    the same generator) on the trainable matrices."""
    import zlib
    g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (1 << 62))
    norm = dim == 1 and ('norm' in name or 'layrnorm' in name) and name.endswith('weight')
    w = (torch.randn(tuple(shape), generator=g) * (0.1 if norm else 0.02) + (1.0 if norm else 0.0)).to(torch.bfloat16)
    if reference and dim >= 2 and trainable_matrix:
        w = (w.to(torch.float32) + 2e-3 * torch.randn(tuple(shape), generator=g)).to(torch.bfloat16)
    return w


class LazyWidthState:
    """Mapping name -> fp32 tensor (bf16-representable values) of llava7b_width's policy or reference model, generated on access: the full-depth fixture's
    27 GB per model never sit in host memory twice.  `materialize()` draws every tensor in a thread pool (each has its own generator, so the values do not
    depend on the order or the thread count) and returns a plain dict of bf16 tensors (lossless)."""

    def __init__(self, shapes, seed, reference):
        self.shapes, self.seed, self.reference = dict(shapes), seed, reference

    def __contains__(self, n):
        return n in self.shapes

    def __iter__(self):
        return iter(self.shapes)

    def __len__(self):
        return len(self.shapes)

    def keys(self):
        return self.shapes.keys()

    def bf16(self, n):
        shape = self.shapes[n]
        return _width_tensor(n, shape, len(shape), self.seed, self.reference, 'vision_tower' not in n)

    def __getitem__(self, n):
        return self.bf16(n).to(torch.float32)

    def materialize(self, workers=16):
        from concurrent.futures import ThreadPoolExecutor
        names = list(self.shapes)
        with ThreadPoolExecutor(workers) as ex:
            return dict(zip(names, ex.map(self.bf16, names)))


def llava7b_width(num_layers=4, T=640, R=48, left_pad=(0, 23), seed=42, lazy=False):
    """One preference pair at the FULL WIDTH of BASELINE.json configs[1] (LLaVA-1.5-7B: CLIP-L/14-336 tower of 24 x 1024, projector, Llama
    layers of 4096 / 11008 / 32 heads x 128, vocabulary 32064) but `num_layers` decoder layers, so that the UNMODIFIED reference trainer runs
    it in fp32 on the build container's CPU in minutes (a full-width parity point that is not HIP-vs-HIP).

    Weights: every tensor of the HF state dict is drawn from its OWN generator, seeded by (seed, crc32 of the parameter name) -- independent of the
    order in which any transformers version initialises its modules, and 10 x faster than an HF init: matrices, embeddings and biases N(0, 0.02),
    norm weights 1 + N(0, 0.1), all rounded to bf16-representable values (the bf16 path and the fp32 twin then load identical numbers);
    reference model = policy + N(0, 2e-3) on the decoder / projector matrices.  Sequence: BOS + 576 image tokens + text, the rejected row
    left-padded; R response tokens.  Returns (LlavaConfig, policy state dict, reference state dict, batch).  Pure torch CPU RNG + the HF config
    classes: regenerated on the GPU box, the 9 GB of weights are not committed (per-tensor checksums are).  lazy=True (the FULL-DEPTH fixture,
    num_layers=32, T=2048, R=512: VERDICT r5 next #1) returns LazyWidthState mappings instead of dicts."""
    from transformers import LlavaForConditionalGeneration
    cfg = llava7b_width_config(num_layers)
    with torch.device('meta'):
        skel = LlavaForConditionalGeneration(cfg)
    shapes = {n: tuple(p.shape) for n, p in skel.named_parameters()}
    sd, ref_sd = LazyWidthState(shapes, seed, False), LazyWidthState(shapes, seed, True)
    if not lazy:
        sd = {n: sd[n] for n in shapes}
        ref_sd = {n: (ref_sd[n] if len(shapes[n]) >= 2 and 'vision_tower' not in n else sd[n]) for n in shapes}
    gb = torch.Generator().manual_seed(seed + 2)
    N = 2
    ids = torch.full((N, T), 32001, dtype=torch.long)
    mask = torch.zeros((N, T), dtype=torch.long)
    prompt = torch.randint(3, 31999, (T - 577 - R,), generator=gb)
    for r in range(N):
        lp = left_pad[r]
        resp = torch.randint(3, 31999, (R - lp,), generator=gb)
        row = torch.cat([torch.tensor([1]), torch.full((576,), 32000), prompt, resp])
        ids[r, lp:] = row
        mask[r, lp:] = 1
    pix = torch.randn(1, 3, 336, 336, generator=gb)
    batch = {'input_ids': ids, 'attention_mask': mask, 'pixel_values': torch.cat([pix, pix], 0),
             'meta_info': {'response_lens': [R - lp for lp in left_pad]}}
    return cfg, sd, ref_sd, batch


def llama31_width_config(num_layers=4):
    """meta-llama/Llama-3.1-8B-Instruct's geometry (the model every scripts/llama/*.sh launcher of the reference loads: llama_dpo.sh:19, llama_ppo.sh:19,
    llama_grpo.sh:19) with `num_layers` decoder layers: hidden 4096, ffn 14336, GQA 32 / 8 x 128, vocabulary 128256, rope theta 500000 with the llama3
    frequency scaling (factor 8, low / high frequency factors 1 / 4, original context 8192), rms eps 1e-5, untied head."""
    from transformers import LlamaConfig
    rp = {'rope_type': 'llama3', 'rope_theta': 500000.0, 'factor': 8.0, 'low_freq_factor': 1.0, 'high_freq_factor': 4.0, 'original_max_position_embeddings': 8192}
    return LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=num_layers, num_attention_heads=32, num_key_value_heads=8, vocab_size=128256,
                       rms_norm_eps=1e-5, max_position_embeddings=131072, rope_parameters=rp, tie_word_embeddings=False, pad_token_id=128004,
                       bos_token_id=128000, eos_token_id=128009, attn_implementation='eager')


def llama31_width(num_layers=4, T=384, R=48, left_pad=(0, 29), seed=43):
    """One preference pair at the FULL WIDTH of the reference's default text backbone (llama31_width_config), built like llava7b_width: every tensor from
    its own generator seeded by (seed, crc32(name)), bf16-representable values, reference model = policy + N(0, 2e-3) on the matrices; the rejected row
    left-padded.  Returns (LlamaConfig, policy state dict, reference state dict, batch).  Regenerated on the GPU box from the seed (per-tensor checksums
    are committed with the fixture, not the 7 GB of weights)."""
    import zlib
    from transformers import LlamaForCausalLM
    cfg = llama31_width_config(num_layers)
    with torch.device('meta'):
        skel = LlamaForCausalLM(cfg)
    sd, ref_sd = {}, {}
    for n, p in skel.named_parameters():
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(n.encode())) % (1 << 62))
        norm = p.dim() == 1 and 'norm' in n
        w = torch.randn(tuple(p.shape), generator=g) * (0.1 if norm else 0.02) + (1.0 if norm else 0.0)
        sd[n] = w.to(torch.bfloat16).to(torch.float32)
        ref_sd[n] = (sd[n] + 2e-3 * torch.randn(tuple(p.shape), generator=g)).to(torch.bfloat16).to(torch.float32) if p.dim() >= 2 else sd[n]
    gb = torch.Generator().manual_seed(seed + 2)
    N, pad = 2, cfg.pad_token_id
    ids = torch.full((N, T), pad, dtype=torch.long)
    mask = torch.zeros((N, T), dtype=torch.long)
    prompt = torch.randint(3, 128000, (T - 1 - R,), generator=gb)
    for r in range(N):
        lp = left_pad[r]
        resp = torch.randint(3, 128000, (R - lp,), generator=gb)
        ids[r, lp:] = torch.cat([torch.tensor([cfg.bos_token_id]), prompt, resp])
        mask[r, lp:] = 1
    return cfg, sd, ref_sd, {'input_ids': ids, 'attention_mask': mask, 'meta_info': {'response_lens': [R - lp for lp in left_pad]}}


def qwen2vl_width_config(num_layers=4, vision_depth=32):
    """Qwen/Qwen2-VL-7B-Instruct's geometry (BASELINE configs[2]) with `num_layers` decoder layers: text 3584 / 18944, GQA 28 / 4 x 128, vocabulary 152064,
    rope theta 1e6 with the multimodal sections [16, 24, 24], rms eps 1e-6; vision tower at FULL depth and width (32 blocks of 1280, 16 heads of 80,
    mlp ratio 4, 14 x 14 x 2 patches, 2 x 2 merger into 3584)."""
    from transformers import Qwen2VLConfig
    return Qwen2VLConfig(
        text_config=dict(hidden_size=3584, intermediate_size=18944, num_hidden_layers=num_layers, num_attention_heads=28, num_key_value_heads=4,
                         vocab_size=152064, max_position_embeddings=32768, rms_norm_eps=1e-6, tie_word_embeddings=False,
                         rope_parameters={'rope_type': 'default', 'rope_theta': 1000000.0, 'mrope_section': [16, 24, 24]}),
        vision_config=dict(depth=vision_depth, embed_dim=1280, hidden_size=3584, num_heads=16, mlp_ratio=4, patch_size=14, temporal_patch_size=2,
                           spatial_merge_size=2, in_channels=3),
        image_token_id=151655, video_token_id=151656, vision_start_token_id=151652, vision_end_token_id=151653, bos_token_id=151643, eos_token_id=151645,
        tie_word_embeddings=False)


def qwen2vl_width(num_layers=4, vision_depth=32, T=320, R=48, left_pad=(0, 17), grid=(1, 16, 16), seed=44):
    """One preference pair at the FULL WIDTH of BASELINE configs[2]'s backbone, built like llava7b_width (per-tensor generators seeded by (seed, crc32(name)),
    bf16-representable values, reference model = policy + N(0, 2e-3) on the decoder / merger matrices).  One image of grid[1] x grid[2] patches
    (-> grid[1] * grid[2] / 4 image tokens) shared by the chosen and the rejected row, as the reference's collator stacks it (images * 2); the rejected row
    is left-padded.  Returns (Qwen2VLConfig, policy state dict, reference state dict, batch)."""
    import zlib
    from transformers import Qwen2VLForConditionalGeneration
    cfg = qwen2vl_width_config(num_layers, vision_depth)
    with torch.device('meta'):
        skel = Qwen2VLForConditionalGeneration(cfg)
    sd, ref_sd = {}, {}
    for n, p in skel.named_parameters():
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(n.encode())) % (1 << 62))
        norm = p.dim() == 1 and ('norm' in n or 'ln_q' in n) and n.endswith('weight')
        w = torch.randn(tuple(p.shape), generator=g) * (0.1 if norm else 0.02) + (1.0 if norm else 0.0)
        sd[n] = w.to(torch.bfloat16).to(torch.float32)
        if p.dim() >= 2 and 'visual.blocks' not in n and 'patch_embed' not in n:
            ref_sd[n] = (sd[n] + 2e-3 * torch.randn(tuple(p.shape), generator=g)).to(torch.bfloat16).to(torch.float32)
        else:
            ref_sd[n] = sd[n]
    gb = torch.Generator().manual_seed(seed + 2)
    PAD, IMG, N = 151643, 151655, 2
    ntok = grid[1] * grid[2] // 4
    ids = torch.full((N, T), PAD, dtype=torch.long)
    mask = torch.zeros((N, T), dtype=torch.long)
    prompt = torch.randint(3, 151000, (T - 3 - ntok - R,), generator=gb)
    for r in range(N):
        lp = left_pad[r]
        resp = torch.randint(3, 151000, (R - lp,), generator=gb)
        ids[r, lp:] = torch.cat([torch.tensor([151644, 151652]), torch.full((ntok,), IMG), torch.tensor([151653]), prompt, resp])      # <|im_start|> <|vision_start|> image <|vision_end|>
        mask[r, lp:] = 1
    pix = torch.randn(grid[0] * grid[1] * grid[2], 3 * 2 * 14 * 14, generator=gb)
    batch = {'input_ids': ids, 'attention_mask': mask, 'pixel_values': torch.cat([pix, pix], 0), 'image_grid_thw': torch.tensor([list(grid), list(grid)]),
             'mm_token_type_ids': (ids == IMG).int(), 'meta_info': {'response_lens': [R - lp for lp in left_pad]}}
    return cfg, sd, ref_sd, batch, PAD


def _seeded_state(skel, seed, frozen_like=(), norm_like=('norm', 'layer_norm', 'ln_')):
    """Every parameter of a meta-device skeleton from its OWN generator (seed, crc32(name)): matrices / embeddings / biases N(0, 0.02), norm weights
    1 + N(0, 0.1), bf16-representable; reference model = policy + N(0, 2e-3) on every matrix whose name holds none of `frozen_like`."""
    import zlib
    sd, ref_sd = {}, {}
    for n, p in skel.named_parameters():
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(n.encode())) % (1 << 62))
        norm = p.dim() == 1 and any(t in n for t in norm_like) and n.endswith('weight')
        w = torch.randn(tuple(p.shape), generator=g) * (0.1 if norm else 0.02) + (1.0 if norm else 0.0)
        sd[n] = w.to(torch.bfloat16).to(torch.float32)
        if p.dim() >= 2 and not any(t in n for t in frozen_like):
            ref_sd[n] = (sd[n] + 2e-3 * torch.randn(tuple(p.shape), generator=g)).to(torch.bfloat16).to(torch.float32)
        else:
            ref_sd[n] = sd[n]
    return sd, ref_sd


def qwen2audio_width_config(num_layers=4, encoder_layers=32):
    """Qwen/Qwen2-Audio-7B-Instruct's geometry (BASELINE configs[3]) with `num_layers` decoder layers: text 4096 / 11008, 32 heads x 128 (MHA), vocabulary
    156032; audio encoder at FULL depth and width (32 layers of 1280, 20 heads of 64, ffn 5120, 128 mel bins, 1500 source positions -> 750 audio tokens per
    30 s clip)."""
    from transformers import Qwen2AudioConfig
    return Qwen2AudioConfig(
        audio_config=dict(num_mel_bins=128, encoder_layers=encoder_layers, encoder_attention_heads=20, encoder_ffn_dim=5120, d_model=1280, max_source_positions=1500),
        text_config=dict(model_type='qwen2', hidden_size=4096, intermediate_size=11008, num_hidden_layers=num_layers, num_attention_heads=32,
                         num_key_value_heads=32, vocab_size=156032, max_position_embeddings=8192, rms_norm_eps=1e-5, rope_theta=10000.0, tie_word_embeddings=False),
        audio_token_id=151646)


def qwen2audio_width(num_layers=4, encoder_layers=32, T=896, R=48, left_pad=(0, 21), frames=3000, seed=45):
    """One preference pair at the FULL WIDTH of BASELINE configs[3]'s backbone (see llava7b_width for the recipe): one 30 s clip (`frames` mel frames -> 750
    audio tokens) shared by the chosen and the rejected row as the reference's collator stacks it; the rejected row left-padded.  Returns
    (Qwen2AudioConfig, policy state dict, reference state dict, batch, pad id)."""
    from transformers import Qwen2AudioForConditionalGeneration
    cfg = qwen2audio_width_config(num_layers, encoder_layers)
    with torch.device('meta'):
        skel = Qwen2AudioForConditionalGeneration(cfg)
    sd, ref_sd = _seeded_state(skel, seed, frozen_like=('embed_positions',))
    gb = torch.Generator().manual_seed(seed + 2)
    PAD, AUD, N = 151643, 151646, 2
    olen = ((frames - 1) // 2 + 1 - 2) // 2 + 1
    ids = torch.full((N, T), PAD, dtype=torch.long)
    mask = torch.zeros((N, T), dtype=torch.long)
    prompt = torch.randint(3, 151000, (T - 2 - olen - R,), generator=gb)
    for r in range(N):
        lp = left_pad[r]
        resp = torch.randint(3, 151000, (R - lp,), generator=gb)
        ids[r, lp:] = torch.cat([torch.tensor([151644]), torch.full((olen,), AUD), torch.tensor([151645]), prompt, resp])
        mask[r, lp:] = 1
    feat = torch.randn(1, 128, 3000, generator=gb)
    fmask = torch.zeros(1, 3000, dtype=torch.long)
    fmask[0, :frames] = 1
    feat[0, :, frames:] = 0.0
    batch = {'input_ids': ids, 'attention_mask': mask, 'input_features': torch.cat([feat, feat], 0), 'feature_attention_mask': torch.cat([fmask, fmask], 0),
             'meta_info': {'response_lens': [R - lp for lp in left_pad]}}
    return cfg, sd, ref_sd, batch, PAD


def qwen3moe_width_config(num_layers=2):
    """Qwen/Qwen3-30B-A3B's layer geometry (BASELINE configs[4]) with `num_layers` sparse layers: hidden 2048, 128 experts of width 768 with top-8 routing
    (normalised), 32 query / 4 kv heads of 128 with per-head q / k norms, vocabulary 151936, rope theta 1e6."""
    from transformers import Qwen3MoeConfig
    return Qwen3MoeConfig(hidden_size=2048, intermediate_size=6144, moe_intermediate_size=768, num_hidden_layers=num_layers, num_attention_heads=32,
                          num_key_value_heads=4, head_dim=128, vocab_size=151936, num_experts=128, num_experts_per_tok=8, norm_topk_prob=True,
                          max_position_embeddings=40960, rms_norm_eps=1e-6, rope_parameters={'rope_type': 'default', 'rope_theta': 1000000.0},
                          tie_word_embeddings=False, pad_token_id=151643)


def qwen3moe_width(num_layers=2, T=320, R=48, left_pad=(0, 19), seed=46):
    """One preference pair at the FULL WIDTH of BASELINE configs[4]'s backbone (all 128 experts per layer; recipe of llava7b_width).  The router matrices are
    drawn wider (N(0, 0.2)) so that the top-8 cut is not a field of near-ties: which experts a token visits is then a property of the model, not of the last
    bit of a dot product.  Returns (Qwen3MoeConfig, policy state dict, reference state dict, batch, pad id)."""
    from transformers import Qwen3MoeForCausalLM
    cfg = qwen3moe_width_config(num_layers)
    with torch.device('meta'):
        skel = Qwen3MoeForCausalLM(cfg)
    sd, ref_sd = _seeded_state(skel, seed)
    for n in sd:
        if n.endswith('mlp.gate.weight'):
            sd[n] = (sd[n] * 10.0).to(torch.bfloat16).to(torch.float32)
            ref_sd[n] = sd[n]                      # the frozen model routes like the policy (its other matrices differ)
    gb = torch.Generator().manual_seed(seed + 2)
    PAD, N = 151643, 2
    ids = torch.full((N, T), PAD, dtype=torch.long)
    mask = torch.zeros((N, T), dtype=torch.long)
    prompt = torch.randint(3, 151000, (T - R,), generator=gb)
    for r in range(N):
        lp = left_pad[r]
        resp = torch.randint(3, 151000, (R - lp,), generator=gb)
        ids[r, lp:] = torch.cat([prompt, resp])
        mask[r, lp:] = 1
    return cfg, sd, ref_sd, {'input_ids': ids, 'attention_mask': mask, 'meta_info': {'response_lens': [R - lp for lp in left_pad]}}, PAD
