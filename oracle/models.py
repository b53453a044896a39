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

import math

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
def llama_decoder(sd, cfg, x, key_valid, prefix='model.language_model.'):
    """hf:models/llama/modeling_llama.py:295-325 (layer), :385-413 (stack).  position_ids = arange(T) for
    every row, i.e. RoPE positions count left-pad tokens (SURVEY.md §8 a')."""
    N, T, h = x.shape
    H, Hkv = cfg['num_heads'], cfg['num_kv_heads']
    hd = cfg['head_dim']
    cos, sin = rope_tables(T, hd, cfg['rope_theta'], x.dtype)
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
