"""TEST INFRASTRUCTURE ONLY.  Functional CPU restatement (plain torch ops, dtype-agnostic, fp32 by
default) of the transformer arithmetic that `model(**batch).logits` executes in the reference hot path
(align_anything/trainers/text_to_text/dpo.py:128).

The reference's model classes are empty subclasses of HuggingFace `transformers` classes
(align_anything/models/llava.py:29, models/opt.py:28), so the arithmetic lives in the third-party
package `transformers` (installed 5.15.0; reference pins >= 4.50.0, pyproject.toml:37).  Citations
prefixed hf: are relative to the installed `transformers/` directory.  Pinned against the HF modules
themselves by tests/test_oracle_golden.py (live, CPU) and tests/golden/models_*.npz.

All functions take a flat state dict with HF key names (what save_pretrained writes) and a plain
config dict (see align_anything_amd/configs.py), and work with autograd so the same code is the
backward oracle.
"""
from __future__ import annotations


import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ shared pieces
def rms_norm(x, w, eps):
    """hf:models/llama/modeling_llama.py:62-67 -- fp32 variance, cast back, then weight multiply."""
    dt = x.dtype
    xf = x.to(torch.float32)
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return w * xf.to(dt)


def rope_tables(T, hd, theta, dtype=torch.float32):
    """hf:models/llama/modeling_llama.py:113-127 -- inv_freq fp32, cos/sin cast to the activation dtype."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.int64).to(torch.float32) / hd))
    pos = torch.arange(T, dtype=torch.float32)
    freqs = pos[:, None] * inv_freq[None, :]
    return freqs.cos().to(dtype), freqs.sin().to(dtype)  # [T, hd/2] (HF concatenates the two halves)


def apply_rope(x, cos, sin):
    """hf:models/llama/modeling_llama.py:130-160 -- half-split rotate_half.  x: [N, H, T, hd]."""
    half = x.shape[-1] // 2
    if cos.dim() == 3:          # per-row tables [N, T, hd/2] (multimodal RoPE)
        c = torch.cat([cos, cos], -1)[:, None]
        s = torch.cat([sin, sin], -1)[:, None]
    else:
        c = torch.cat([cos, cos], -1)[None, None]
        s = torch.cat([sin, sin], -1)[None, None]
    rot = torch.cat([-x[..., half:], x[..., :half]], -1)
    return x * c + rot * s


def attention(q, k, v, scale, causal, key_valid=None):
    """Eager attention (hf:models/llama/modeling_llama.py:179-214): fp32 softmax of scaled scores with an
    additive mask (causal + key padding), probabilities cast back to the value dtype.
    q,k,v: [N, H, T, hd]; key_valid: bool [N, T]."""
    N, H, T, hd = q.shape
    scores = torch.matmul(q, k.transpose(-1, -2)) * scale
    mask = torch.zeros((N, 1, T, T), dtype=torch.bool)
    if causal:
        mask = mask | ~torch.tril(torch.ones(T, T, dtype=torch.bool))[None, None]
    if key_valid is not None:
        mask = mask | ~key_valid[:, None, None, :]
    scores = scores.masked_fill(mask, torch.finfo(scores.dtype).min)
    p = F.softmax(scores, dim=-1, dtype=torch.float32).to(q.dtype)
    return torch.matmul(p, v)


def linear(x, sd, prefix):
    return F.linear(x, sd[prefix + '.weight'], sd.get(prefix + '.bias'))


# ------------------------------------------------------------------ Llama decoder
def llama_decoder(sd, cfg, x, key_valid, prefix='model.language_model.', cos_sin=None):
    """hf:models/llama/modeling_llama.py:295-325 (layer), :385-413 (stack).  position_ids = arange(T) for
    every row, i.e. RoPE positions count left-pad tokens (SURVEY.md §8 a')."""
    N, T, h = x.shape
    H, Hkv = cfg['num_heads'], cfg['num_kv_heads']
    hd = cfg['head_dim']
    cos, sin = rope_tables(T, hd, cfg['rope_theta'], x.dtype) if cos_sin is None else cos_sin
    for i in range(cfg['num_layers']):
        p = f'{prefix}layers.{i}.'
        r = x
        y = rms_norm(x, sd[p + 'input_layernorm.weight'], cfg['rms_eps'])
        q = linear(y, sd, p + 'self_attn.q_proj').view(N, T, H, hd).transpose(1, 2)
        k = linear(y, sd, p + 'self_attn.k_proj').view(N, T, Hkv, hd).transpose(1, 2)
        v = linear(y, sd, p + 'self_attn.v_proj').view(N, T, Hkv, hd).transpose(1, 2)
        q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
        if Hkv != H:
            k = k.repeat_interleave(H // Hkv, dim=1)
            v = v.repeat_interleave(H // Hkv, dim=1)
        a = attention(q, k, v, hd ** -0.5, True, key_valid)
        a = a.transpose(1, 2).reshape(N, T, H * hd)
        x = r + linear(a, sd, p + 'self_attn.o_proj')
        r = x
        y = rms_norm(x, sd[p + 'post_attention_layernorm.weight'], cfg['rms_eps'])
        g = linear(y, sd, p + 'mlp.gate_proj')
        u = linear(y, sd, p + 'mlp.up_proj')
        x = r + linear(F.silu(g) * u, sd, p + 'mlp.down_proj')
    return rms_norm(x, sd[prefix + 'norm.weight'], cfg['rms_eps'])


# ------------------------------------------------------------------ CLIP vision tower
def clip_vision(sd, vcfg, pixel_values, prefix='model.vision_tower.'):
    """hf:models/clip/modeling_clip.py:138-218 (embeddings), :338-351 (layer), :605-651 (model).
    Returns the list of hidden states (index 0 = embeddings after pre_layrnorm)."""
    N = pixel_values.shape[0]
    h, P = vcfg['hidden_size'], vcfg['patch_size']
    w = sd[prefix + 'embeddings.patch_embedding.weight']
    pe = F.conv2d(pixel_values.to(w.dtype), w, stride=P).flatten(2).transpose(1, 2)  # [N, G*G, h]
    cls = sd[prefix + 'embeddings.class_embedding'].expand(N, 1, -1)
    x = torch.cat([cls, pe], 1) + sd[prefix + 'embeddings.position_embedding.weight'][None]
    x = F.layer_norm(x, (h,), sd[prefix + 'pre_layrnorm.weight'], sd[prefix + 'pre_layrnorm.bias'], vcfg['ln_eps'])
    hs = [x]
    H = vcfg['num_heads']
    hd = h // H
    T = x.shape[1]
    for i in range(vcfg['num_layers']):
        p = f'{prefix}encoder.layers.{i}.'
        r = x
        y = F.layer_norm(x, (h,), sd[p + 'layer_norm1.weight'], sd[p + 'layer_norm1.bias'], vcfg['ln_eps'])
        q = linear(y, sd, p + 'self_attn.q_proj').view(N, T, H, hd).transpose(1, 2)
        k = linear(y, sd, p + 'self_attn.k_proj').view(N, T, H, hd).transpose(1, 2)
        v = linear(y, sd, p + 'self_attn.v_proj').view(N, T, H, hd).transpose(1, 2)
        a = attention(q, k, v, hd ** -0.5, False).transpose(1, 2).reshape(N, T, h)
        x = r + linear(a, sd, p + 'self_attn.out_proj')
        r = x
        y = F.layer_norm(x, (h,), sd[p + 'layer_norm2.weight'], sd[p + 'layer_norm2.bias'], vcfg['ln_eps'])
        y = linear(y, sd, p + 'mlp.fc1')
        y = y * torch.sigmoid(1.702 * y)  # quick_gelu
        x = r + linear(y, sd, p + 'mlp.fc2')
        hs.append(x)
    return hs


# ------------------------------------------------------------------ LLaVA
def llava_hidden(sd, cfg, input_ids, attention_mask, pixel_values):
    """hf:models/llava/modeling_llava.py:154-166 (image features: hidden_states[-2][:, 1:]), :87-106
    (projector), :234-248 (masked_scatter merge), then the Llama stack.  Returns final-norm hidden."""
    emb = sd['model.language_model.embed_tokens.weight']
    x = F.embedding(input_ids, emb)
    if pixel_values is not None:
        hs = clip_vision(sd, cfg['vision'], pixel_values)
        feat = hs[cfg['vision_feature_layer']][:, 1:]
        feat = linear(feat, sd, 'model.multi_modal_projector.linear_1')
        feat = F.gelu(feat)
        feat = linear(feat, sd, 'model.multi_modal_projector.linear_2')
        mask = (input_ids == cfg['image_token_id'])
        assert int(mask.sum()) == feat.shape[0] * feat.shape[1], 'image token / feature count mismatch'
        x = x.masked_scatter(mask[..., None].expand_as(x), feat.to(x.dtype))
    key_valid = attention_mask.bool() if attention_mask is not None else None
    return llama_decoder(sd, cfg['text'], x, key_valid)


def llava_logits(sd, cfg, input_ids, attention_mask, pixel_values):
    """hf:models/llava/modeling_llava.py:359-361 -- lm_head on ALL positions."""
    return F.linear(llava_hidden(sd, cfg, input_ids, attention_mask, pixel_values), sd['lm_head.weight'])


# ------------------------------------------------------------------ Qwen2-VL
def qwen2vl_vision(sd, vcfg, pixel_values, grid_thw, prefix='model.visual.'):
    """hf:models/qwen2_vl/modeling_qwen2_vl.py:251-291 (patch embed = Conv3d with kernel == stride, i.e. a matmul over the
    flattened patches; merger), :342-450 (block: LayerNorm, fused qkv, 2-D rotary in fp32, full attention within each
    temporal frame, quick_gelu MLP), :700-730 (model); hf:vision_utils.py get_vision_position_ids (block-major h/w ids).
    pixel_values: [n_patches, C*tps*ps*ps]; grid_thw: [[t, h, w], ...].  Returns merged features [n_patches/merge^2, out]."""
    E, H, m = vcfg['embed_dim'], vcfg['num_heads'], vcfg['spatial_merge_size']
    hd = E // H
    w = sd[prefix + 'patch_embed.proj.weight']
    x = F.linear(pixel_values.to(w.dtype), w.reshape(E, -1))
    pos = []
    for t, h, wd in grid_thw:
        hp, wp = torch.meshgrid(torch.arange(h), torch.arange(wd), indexing='ij')
        blk = (h // m, m, wd // m, m)
        hp = hp.reshape(blk).transpose(1, 2).flatten()
        wp = wp.reshape(blk).transpose(1, 2).flatten()
        pos.append(torch.stack([hp, wp], -1).repeat(t, 1))
    pos = torch.cat(pos, 0)
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float) / (hd // 2)))
    freqs = (pos.unsqueeze(-1) * inv_freq).flatten(1)               # [n, hd/2]
    emb = torch.cat([freqs, freqs], -1)
    cos, sin = emb.cos()[:, None].float(), emb.sin()[:, None].float()

    def rot(t):                                                      # apply_rotary_pos_emb_vision: fp32, cast back
        tf = t.float()
        half = tf.shape[-1] // 2
        r = torch.cat([-tf[..., half:], tf[..., :half]], -1)
        return (tf * cos + r * sin).to(t.dtype)

    segs = [h * wd for t, h, wd in grid_thw for _ in range(t)]
    n = x.shape[0]
    for i in range(vcfg['depth']):
        p = f'{prefix}blocks.{i}.'
        y = F.layer_norm(x, (E,), sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], 1e-6)
        q, k, v = linear(y, sd, p + 'attn.qkv').reshape(n, 3, H, hd).permute(1, 0, 2, 3).unbind(0)
        q, k = rot(q), rot(k)
        outs, o = [], 0
        for L in segs:                                               # one attention segment per (image, frame)
            a = attention(q[o:o + L].transpose(0, 1)[None], k[o:o + L].transpose(0, 1)[None], v[o:o + L].transpose(0, 1)[None],
                          hd ** -0.5, False)
            outs.append(a[0].transpose(0, 1).reshape(L, E)); o += L
        x = x + linear(torch.cat(outs, 0), sd, p + 'attn.proj')
        y = F.layer_norm(x, (E,), sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], 1e-6)
        y = linear(y, sd, p + 'mlp.fc1')
        x = x + linear(y * torch.sigmoid(1.702 * y), sd, p + 'mlp.fc2')
    y = F.layer_norm(x, (E,), sd[prefix + 'merger.ln_q.weight'], sd[prefix + 'merger.ln_q.bias'], 1e-6).view(-1, E * m * m)
    return linear(F.gelu(linear(y, sd, prefix + 'merger.mlp.0')), sd, prefix + 'merger.mlp.2')


def qwen2vl_rope_index(input_ids, attention_mask, grid_thw, image_token_id, merge):
    """hf:models/qwen2_vl/modeling_qwen2_vl.py:862-1018 (get_vision_position_ids + get_rope_index) with the token types
    taken from input_ids == image_token_id (what the processor's mm_token_type_ids encodes).  Integer work: must match
    HF bit-exactly.  Returns position_ids int64 [3, N, T] (pad positions 0) and the per-row rope deltas."""
    N, T = input_ids.shape
    pos = torch.zeros(3, N, T, dtype=torch.int64)
    grids = iter(grid_thw)
    deltas = []
    for b in range(N):
        keep = attention_mask[b].bool() if attention_mask is not None else torch.ones(T, dtype=torch.bool)
        types = (input_ids[b][keep] == image_token_id).tolist()
        cur, chunks, i = 0, [], 0
        while i < len(types):
            j = i
            while j < len(types) and types[j] == types[i]:
                j += 1
            if not types[i]:
                chunks.append(torch.arange(j - i).view(1, -1).expand(3, -1) + cur)
                cur += j - i
            else:
                t, h, w = next(grids)
                gh, gw = h // merge, w // merge
                tt, hh, ww = torch.meshgrid(torch.arange(t), torch.arange(gh) + cur, torch.arange(gw) + cur, indexing='ij')
                v = torch.stack([tt, hh, ww], 0).reshape(3, -1)
                v[0] += cur
                chunks.append(v)
                cur += max(h, w) // merge
            i = j
        lp = torch.cat(chunks, 1)
        pos[:, b, keep] = lp
        deltas.append(int(lp.max()) + 1 - int(keep.sum()))
    return pos, torch.tensor(deltas)


def qwen2vl_hidden(sd, cfg, input_ids, attention_mask, pixel_values, grid_thw):
    """hf:models/qwen2_vl/modeling_qwen2_vl.py:1144-1205 (merge image features by masked_scatter, 3-D position ids) and
    :156-222 (multimodal RoPE: frequency f of the half-dim takes its angle from the t / h / w position according to
    mrope_section), then the Qwen2 decoder (= the Llama block with q/k/v biases)."""
    t = cfg['text']
    x = F.embedding(input_ids, sd['model.language_model.embed_tokens.weight'])
    if pixel_values is not None:
        feat = qwen2vl_vision(sd, cfg['vision'], pixel_values, grid_thw)
        mask = input_ids == cfg['image_token_id']
        assert int(mask.sum()) == feat.shape[0], 'image token / feature count mismatch'
        x = x.masked_scatter(mask[..., None].expand_as(x), feat.to(x.dtype))
        pos, _ = qwen2vl_rope_index(input_ids, attention_mask, grid_thw, cfg['image_token_id'], cfg['vision']['spatial_merge_size'])
    else:   # text only: HF derives 1-D positions from the mask, identical on the three axes
        am = attention_mask if attention_mask is not None else torch.ones_like(input_ids)
        p1 = (am.long().cumsum(-1) - 1).masked_fill(am == 0, 0)
        pos = p1[None].expand(3, -1, -1)
    hd = t['head_dim']
    inv_freq = 1.0 / (t['rope_theta'] ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
    freqs = pos[..., None].float() * inv_freq                        # [3, N, T, hd/2]
    sec = t['mrope_section']
    comp = torch.cat([torch.full((n,), i % 3) for i, n in enumerate(sec)])
    f = torch.stack([freqs[int(comp[j]), :, :, j] for j in range(hd // 2)], -1)    # [N, T, hd/2]
    cos, sin = f.cos().to(x.dtype), f.sin().to(x.dtype)
    key_valid = attention_mask.bool() if attention_mask is not None else None
    return llama_decoder(sd, t, x, key_valid, cos_sin=(cos, sin))


def qwen2vl_logits(sd, cfg, input_ids, attention_mask, pixel_values, grid_thw):
    return F.linear(qwen2vl_hidden(sd, cfg, input_ids, attention_mask, pixel_values, grid_thw), sd['lm_head.weight'])


# ------------------------------------------------------------------ Qwen2-Audio
def qwen2audio_lengths(feature_lengths):
    """hf:models/qwen2_audio/modeling_qwen2_audio.py:400-406: frames after conv2 (stride 2) and after AvgPool1d(2)."""
    a = (feature_lengths - 1) // 2 + 1
    return a, (a - 2) // 2 + 1


def qwen2audio_tower(sd, acfg, input_features, feature_lengths, prefix='model.audio_tower.'):
    """hf:models/qwen2_audio/modeling_qwen2_audio.py:343-395 Qwen2AudioEncoder: conv1 + GELU, conv2 (stride 2) + GELU, + sinusoidal
    positions, pre-LN layers (q scaled by head_dim**-0.5, k_proj without bias, keys beyond the audio's length masked), AvgPool1d(2),
    LayerNorm.  input_features [B, mel, 2 * max_source_positions]; returns [B, max_source_positions / 2, d_model]."""
    d, H = acfg['d_model'], acfg['num_heads']
    hd = d // H
    x = F.gelu(F.conv1d(input_features.to(sd[prefix + 'conv1.weight'].dtype), sd[prefix + 'conv1.weight'], sd[prefix + 'conv1.bias'], padding=1))
    x = F.gelu(F.conv1d(x, sd[prefix + 'conv2.weight'], sd[prefix + 'conv2.bias'], stride=2, padding=1))
    x = x.permute(0, 2, 1) + sd[prefix + 'embed_positions.weight']
    B, T, _ = x.shape
    alen, _ = qwen2audio_lengths(feature_lengths)
    key_valid = torch.arange(T)[None, :] < alen[:, None]
    for i in range(acfg['num_layers']):
        p = f'{prefix}layers.{i}.'
        y = F.layer_norm(x, (d,), sd[p + 'self_attn_layer_norm.weight'], sd[p + 'self_attn_layer_norm.bias'], 1e-5)
        q = (linear(y, sd, p + 'self_attn.q_proj') * hd ** -0.5).view(B, T, H, hd).transpose(1, 2)
        k = linear(y, sd, p + 'self_attn.k_proj').view(B, T, H, hd).transpose(1, 2)
        v = linear(y, sd, p + 'self_attn.v_proj').view(B, T, H, hd).transpose(1, 2)
        a = attention(q, k, v, 1.0, False, key_valid).transpose(1, 2).reshape(B, T, d)
        x = x + linear(a, sd, p + 'self_attn.out_proj')
        y = F.layer_norm(x, (d,), sd[p + 'final_layer_norm.weight'], sd[p + 'final_layer_norm.bias'], 1e-5)
        x = x + linear(F.gelu(linear(y, sd, p + 'fc1')), sd, p + 'fc2')
    x = F.avg_pool1d(x.permute(0, 2, 1), 2, stride=2).permute(0, 2, 1)
    return F.layer_norm(x, (d,), sd[prefix + 'layer_norm.weight'], sd[prefix + 'layer_norm.bias'], 1e-5)


def qwen2audio_logits(sd, cfg, input_ids, attention_mask, input_features, feature_attention_mask):
    """hf:models/qwen2_audio/modeling_qwen2_audio.py:674-745 (processor-expanded audio tokens: tower -> Linear projector -> the
    first `output_length` frames of every audio are masked_scatter'ed over the audio-token positions), then the Qwen2
    decoder with 1-D RoPE positions arange(T)."""
    x = F.embedding(input_ids, sd['model.language_model.embed_tokens.weight'])
    if input_features is not None:
        flen = feature_attention_mask.sum(-1)
        feat = linear(qwen2audio_tower(sd, cfg['audio'], input_features, flen), sd, 'model.multi_modal_projector.linear')
        _, olen = qwen2audio_lengths(flen)
        keep = torch.arange(feat.shape[1])[None, :] < olen[:, None]
        feat = feat[keep]
        mask = input_ids == cfg['audio_token_id']
        assert int(mask.sum()) == feat.shape[0], 'audio token / feature count mismatch'
        x = x.masked_scatter(mask[..., None].expand_as(x), feat.to(x.dtype))
    key_valid = attention_mask.bool() if attention_mask is not None else None
    return F.linear(llama_decoder(sd, cfg['text'], x, key_valid), sd['lm_head.weight'])


# ------------------------------------------------------------------ Qwen3-MoE
def qwen3moe_logits(sd, cfg, input_ids, attention_mask, prefix='model.'):
    """hf:models/qwen3_moe/modeling_qwen3_moe.py: attention with per-head RMSNorm on q and k before RoPE (:122-190), sparse MoE
    block (:210-283: fp32 router softmax, top-k, renormalisation, weights cast to the activation dtype, per-expert SwiGLU on the
    3-D `gate_up_proj` / `down_proj` parameters, index_add combine), 1-D RoPE positions arange(T)."""
    N, T = input_ids.shape
    H, Hkv, hd, eps = cfg['num_heads'], cfg['num_kv_heads'], cfg['head_dim'], cfg['rms_eps']
    E, k = cfg['num_experts'], cfg['num_experts_per_tok']
    x = F.embedding(input_ids, sd[prefix + 'embed_tokens.weight'])
    key_valid = attention_mask.bool() if attention_mask is not None else None
    cos, sin = rope_tables(T, hd, cfg['rope_theta'], x.dtype)
    for i in range(cfg['num_layers']):
        p = f'{prefix}layers.{i}.'
        y = rms_norm(x, sd[p + 'input_layernorm.weight'], eps)
        q = rms_norm(linear(y, sd, p + 'self_attn.q_proj').view(N, T, H, hd), sd[p + 'self_attn.q_norm.weight'], eps).transpose(1, 2)
        kk = rms_norm(linear(y, sd, p + 'self_attn.k_proj').view(N, T, Hkv, hd), sd[p + 'self_attn.k_norm.weight'], eps).transpose(1, 2)
        v = linear(y, sd, p + 'self_attn.v_proj').view(N, T, Hkv, hd).transpose(1, 2)
        q, kk = apply_rope(q, cos, sin), apply_rope(kk, cos, sin)
        if Hkv != H:
            kk = kk.repeat_interleave(H // Hkv, dim=1); v = v.repeat_interleave(H // Hkv, dim=1)
        a = attention(q, kk, v, hd ** -0.5, True, key_valid).transpose(1, 2).reshape(N, T, H * hd)
        x = x + linear(a, sd, p + 'self_attn.o_proj')
        y = rms_norm(x, sd[p + 'post_attention_layernorm.weight'], eps).reshape(N * T, -1)
        logits = F.linear(y, sd[p + 'mlp.gate.weight'])
        probs = F.softmax(logits, dim=-1, dtype=torch.float)
        top_w, top_i = torch.topk(probs, k, dim=-1)
        if cfg['norm_topk_prob']:
            top_w = top_w / top_w.sum(-1, keepdim=True)
        top_w = top_w.to(logits.dtype)
        out = torch.zeros_like(y)
        for e in range(E):
            slot, tok = torch.where((top_i == e).t())
            if tok.numel() == 0:
                continue
            g, u = F.linear(y[tok], sd[p + 'mlp.experts.gate_up_proj'][e]).chunk(2, dim=-1)
            h_e = F.linear(F.silu(g) * u, sd[p + 'mlp.experts.down_proj'][e]) * top_w[tok, slot, None]
            out = out.index_add(0, tok, h_e.to(out.dtype))
        x = x + out.view(N, T, -1)
    return F.linear(rms_norm(x, sd[prefix + 'norm.weight'], eps), sd['lm_head.weight'])


def llama_logits(sd, cfg, input_ids, attention_mask, prefix='model.'):
    x = F.embedding(input_ids, sd[prefix + 'embed_tokens.weight'])
    key_valid = attention_mask.bool() if attention_mask is not None else None
    return F.linear(llama_decoder(sd, cfg, x, key_valid, prefix), sd['lm_head.weight'])


# ------------------------------------------------------------------ OPT
def opt_logits(sd, cfg, input_ids, attention_mask, dropout_p=0.0, return_hidden=False):
    """hf:models/opt/modeling_opt.py:45-70 (learned positions = cumsum(mask)*mask - 1 + 2), :191-251
    (pre-LN decoder layer, biased projections, ReLU MLP), :283-310, lm_head tied to embed_tokens.
    Dropout must be 0 for parity (SURVEY.md §7 hard parts)."""
    assert dropout_p == 0.0
    N, T = input_ids.shape
    h, H = cfg['hidden_size'], cfg['num_heads']
    hd = h // H
    emb = sd['model.decoder.embed_tokens.weight']
    am = attention_mask.long() if attention_mask is not None else torch.ones_like(input_ids)
    pos = (torch.cumsum(am, dim=1) * am - 1).long() + 2
    x = F.embedding(input_ids, emb) + F.embedding(pos, sd['model.decoder.embed_positions.weight'])
    key_valid = am.bool()
    for i in range(cfg['num_layers']):
        p = f'model.decoder.layers.{i}.'
        r = x
        y = F.layer_norm(x, (h,), sd[p + 'self_attn_layer_norm.weight'], sd[p + 'self_attn_layer_norm.bias'], 1e-5)
        q = linear(y, sd, p + 'self_attn.q_proj').view(N, T, H, hd).transpose(1, 2)
        k = linear(y, sd, p + 'self_attn.k_proj').view(N, T, H, hd).transpose(1, 2)
        v = linear(y, sd, p + 'self_attn.v_proj').view(N, T, H, hd).transpose(1, 2)
        a = attention(q, k, v, hd ** -0.5, True, key_valid).transpose(1, 2).reshape(N, T, h)
        x = r + linear(a, sd, p + 'self_attn.out_proj')
        r = x
        y = F.layer_norm(x, (h,), sd[p + 'final_layer_norm.weight'], sd[p + 'final_layer_norm.bias'], 1e-5)
        y = F.relu(linear(y, sd, p + 'fc1'))
        x = r + linear(y, sd, p + 'fc2')
    x = F.layer_norm(x, (h,), sd['model.decoder.final_layer_norm.weight'], sd['model.decoder.final_layer_norm.bias'], 1e-5)
    if return_hidden:
        return x
    return F.linear(x, sd.get('lm_head.weight', emb))


# ------------------------------------------------------------------ score head (reward / critic)
def score_from_hidden(hidden, score_w, attention_mask=None, end_at_last_position=True):
    """align_anything/models/llava.py:60-68 (end score = position -1) and models/opt.py:59-89
    (end score at the last attended token).  hidden [N,T,h], score_w [1,h] -> scores [N,T,1], end [N,1]."""
    scores = F.linear(hidden, score_w)
    if end_at_last_position or attention_mask is None:
        end = scores[:, -1]
    else:
        idx = torch.stack([m.nonzero()[-1].squeeze() for m in attention_mask])
        end = scores[torch.arange(scores.shape[0]), idx]
    return scores, end
