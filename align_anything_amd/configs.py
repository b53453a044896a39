"""Plain-dict model geometry for the native path (no HF objects on the hot path).

`from_hf_config` converts a HuggingFace config (what `AnyModel.from_pretrained` reads from config.json,
align_anything/models/model_registry.py:134-175) into the dicts the native modules and the oracle use.
The named constructors give the BASELINE.json geometries without touching the network.
"""
from __future__ import annotations


def rope_scaling_of(c) -> dict | None:
    """The RoPE frequency scaling of an HF (text) config, from `rope_parameters` (transformers >= 5) or `rope_scaling` (4.x; key 'rope_type' or
    'type').  None for the plain rotary embedding (and for M-RoPE, whose sections are read elsewhere).  Built: 'linear' (position interpolation)
    and 'llama3' (Llama-3.1's frequency-dependent scaling: meta-llama/Llama-3.1-8B-Instruct is the reference's text-to-text default,
    scripts/llama/*.sh).  Anything else RAISES: a scaling that is silently dropped gives wrong logits at every position."""
    rp = getattr(c, 'rope_parameters', None) or getattr(c, 'rope_scaling', None) or {}
    kind = rp.get('rope_type', rp.get('type', 'default')) or 'default'
    if kind in ('default', 'mrope'):
        return None
    if kind == 'linear':
        return {'type': 'linear', 'factor': float(rp['factor'])}
    if kind == 'llama3':
        return {'type': 'llama3', 'factor': float(rp['factor']), 'low_freq_factor': float(rp['low_freq_factor']),
                'high_freq_factor': float(rp['high_freq_factor']),
                'original_max_position_embeddings': int(rp.get('original_max_position_embeddings') or c.max_position_embeddings)}
    raise ValueError(f'rope scaling type {kind!r} has no native implementation (default, linear and llama3 are built)')


def check_llama_family(c, what: str) -> None:
    """Options of an HF Llama-family text config that the native decoder does not implement must not be dropped silently."""
    if getattr(c, 'mlp_bias', False):
        raise ValueError(f'{what}: mlp_bias has no native implementation')
    if getattr(c, 'hidden_act', 'silu') not in ('silu', 'swish'):
        raise ValueError(f'{what}: hidden_act {c.hidden_act!r} has no native implementation (silu is built)')
    if float(getattr(c, 'partial_rotary_factor', 1.0) or 1.0) != 1.0 or float((getattr(c, 'rope_parameters', None) or {}).get('partial_rotary_factor', 1.0) or 1.0) != 1.0:
        raise ValueError(f'{what}: partial rotary embeddings have no native implementation')


def llama_cfg(hidden_size, intermediate_size, num_layers, num_heads, num_kv_heads, vocab_size,
              rms_eps=1e-5, rope_theta=10000.0, head_dim=None, max_position_embeddings=4096, attention_bias=False, rope_scaling=None,
              tie_word_embeddings=False):
    return {
        'attention_bias': attention_bias, 'rope_scaling': rope_scaling, 'tie_word_embeddings': bool(tie_word_embeddings),
        'kind': 'llama', 'hidden_size': hidden_size, 'intermediate_size': intermediate_size,
        'num_layers': num_layers, 'num_heads': num_heads, 'num_kv_heads': num_kv_heads,
        'head_dim': head_dim or hidden_size // num_heads, 'vocab_size': vocab_size, 'rms_eps': rms_eps,
        'rope_theta': rope_theta, 'max_position_embeddings': max_position_embeddings,
    }


def clip_vision_cfg(hidden_size, intermediate_size, num_layers, num_heads, image_size, patch_size,
                    ln_eps=1e-5, num_channels=3):
    return {
        'kind': 'clip_vision', 'hidden_size': hidden_size, 'intermediate_size': intermediate_size,
        'num_layers': num_layers, 'num_heads': num_heads, 'image_size': image_size,
        'patch_size': patch_size, 'ln_eps': ln_eps, 'num_channels': num_channels,
    }


def llava_cfg(text, vision, image_token_id, vision_feature_layer=-2, pad_token_id=None):
    return {'kind': 'llava', 'text': text, 'vision': vision, 'image_token_id': image_token_id,
            'vision_feature_layer': vision_feature_layer, 'pad_token_id': pad_token_id}


def qwen2vl_vision_cfg(embed_dim, depth, num_heads, mlp_ratio, hidden_size, patch_size=14, temporal_patch_size=2,
                       spatial_merge_size=2, in_channels=3):
    return {'kind': 'qwen2vl_vision', 'embed_dim': embed_dim, 'depth': depth, 'num_heads': num_heads, 'mlp_ratio': mlp_ratio,
            'hidden_size': hidden_size, 'patch_size': patch_size, 'temporal_patch_size': temporal_patch_size,
            'spatial_merge_size': spatial_merge_size, 'in_channels': in_channels}


def qwen2vl_cfg(text, vision, image_token_id, mrope_section, pad_token_id=None):
    """text = llama_cfg(..., attention_bias=True) (the Qwen2 decoder) + the multimodal-RoPE split of head_dim/2."""
    if sum(mrope_section) != text['head_dim'] // 2:
        raise ValueError(f"mrope_section {mrope_section} must sum to head_dim/2 = {text['head_dim'] // 2}")
    text = dict(text, mrope_section=list(mrope_section))
    return {'kind': 'qwen2vl', 'text': text, 'vision': vision, 'image_token_id': image_token_id, 'pad_token_id': pad_token_id}


def qwen2audio_tower_cfg(d_model, num_layers, num_heads, ffn_dim, num_mel_bins=128, max_source_positions=1500):
    return {'kind': 'qwen2audio_tower', 'd_model': d_model, 'num_layers': num_layers, 'num_heads': num_heads, 'ffn_dim': ffn_dim,
            'num_mel_bins': num_mel_bins, 'max_source_positions': max_source_positions}


def qwen2audio_cfg(text, audio, audio_token_id, pad_token_id=None):
    return {'kind': 'qwen2audio', 'text': text, 'audio': audio, 'audio_token_id': audio_token_id, 'pad_token_id': pad_token_id}


def qwen3moe_cfg(hidden_size, moe_intermediate_size, num_layers, num_heads, num_kv_heads, vocab_size, num_experts, num_experts_per_tok,
                 head_dim=128, norm_topk_prob=True, rms_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=40960):
    """Qwen3-MoE text decoder (every layer sparse): q/k RMSNorm per head, no attention bias, `num_experts` SwiGLU experts of
    width `moe_intermediate_size`, top-`num_experts_per_tok` routing."""
    return {'kind': 'qwen3moe', 'hidden_size': hidden_size, 'moe_intermediate_size': moe_intermediate_size, 'num_layers': num_layers,
            'num_heads': num_heads, 'num_kv_heads': num_kv_heads, 'head_dim': head_dim, 'vocab_size': vocab_size,
            'num_experts': num_experts, 'num_experts_per_tok': num_experts_per_tok, 'norm_topk_prob': bool(norm_topk_prob),
            'rms_eps': rms_eps, 'rope_theta': rope_theta, 'max_position_embeddings': max_position_embeddings}


def opt_cfg(hidden_size, ffn_dim, num_layers, num_heads, vocab_size, max_position_embeddings=2048):
    return {'kind': 'opt', 'hidden_size': hidden_size, 'ffn_dim': ffn_dim, 'num_layers': num_layers,
            'num_heads': num_heads, 'vocab_size': vocab_size,
            'max_position_embeddings': max_position_embeddings}


def llava_1_5_7b(num_layers=32, vision_layers=24):
    """LLaVA-1.5-7B geometry (BASELINE.json config 2): CLIP-L/14-336 + Vicuna-7B, V = 32064."""
    text = llama_cfg(4096, 11008, num_layers, 32, 32, 32064, rms_eps=1e-5)
    vision = clip_vision_cfg(1024, 4096, vision_layers, 16, 336, 14)
    return llava_cfg(text, vision, image_token_id=32000, pad_token_id=32001)


def qwen2_vl_7b(num_layers=28, vision_depth=32):
    """Qwen2-VL-7B-Instruct geometry (BASELINE.json config 3): ViT 1280 x 32 (16 heads of 80) + Qwen2-7B (28 x 3584, GQA 28/4,
    ffn 18944, V = 152064), mrope_section [16, 24, 24], theta 1e6."""
    text = llama_cfg(3584, 18944, num_layers, 28, 4, 152064, rms_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=32768,
                     attention_bias=True)
    vision = qwen2vl_vision_cfg(1280, vision_depth, 16, 4, 3584)
    return qwen2vl_cfg(text, vision, image_token_id=151655, mrope_section=[16, 24, 24], pad_token_id=151643)


def qwen2_audio_7b(num_layers=32, audio_layers=32):
    """Qwen2-Audio-7B-Instruct geometry (BASELINE.json config 4): Whisper-large-v3 style encoder (1280 x 32, 20 heads of 64, ffn 5120,
    128 mel bins, 1500 positions -> 750 audio tokens per 30 s) + Qwen-7B decoder (32 x 4096, MHA 32, ffn 11008, V = 156032, q/k/v bias)."""
    text = llama_cfg(4096, 11008, num_layers, 32, 32, 156032, rms_eps=1e-5, rope_theta=10000.0, max_position_embeddings=8192, attention_bias=True)
    audio = qwen2audio_tower_cfg(1280, audio_layers, 20, 5120, num_mel_bins=128, max_source_positions=1500)
    return qwen2audio_cfg(text, audio, audio_token_id=151646, pad_token_id=151643)


def opt_125m():
    """facebook/opt-125m geometry (BASELINE.json config 1), dropout forced to 0 for parity."""
    return opt_cfg(768, 3072, 12, 12, 50272, 2048)


def from_hf_config(c) -> dict:
    mt = getattr(c, 'model_type', None)
    if mt == 'llava':
        t, v = c.text_config, c.vision_config
        rp = getattr(t, 'rope_parameters', None) or {}
        theta = rp.get('rope_theta', getattr(t, 'rope_theta', 10000.0))
        check_llama_family(t, 'llava text_config')
        if getattr(v, 'model_type', 'clip_vision_model') != 'clip_vision_model' or getattr(v, 'hidden_act', 'quick_gelu') != 'quick_gelu':
            raise ValueError(f'llava: vision tower {getattr(v, "model_type", None)!r} with activation {getattr(v, "hidden_act", None)!r} has no native implementation '
                             '(the CLIP tower with quick_gelu is built)')
        if getattr(c, 'vision_feature_select_strategy', 'default') != 'default' or getattr(c, 'projector_hidden_act', 'gelu') != 'gelu' \
                or not getattr(c, 'multimodal_projector_bias', True) or isinstance(c.vision_feature_layer, (list, tuple)):
            raise ValueError('llava: vision_feature_select_strategy / projector_hidden_act / multimodal_projector_bias / a list of feature layers other than '
                             'the LLaVA-1.5 defaults (default, gelu, biased, one layer) have no native implementation')
        text = llama_cfg(t.hidden_size, t.intermediate_size, t.num_hidden_layers, t.num_attention_heads,
                         t.num_key_value_heads, t.vocab_size, t.rms_norm_eps, theta,
                         getattr(t, 'head_dim', None), t.max_position_embeddings, attention_bias=bool(getattr(t, 'attention_bias', False)),
                         rope_scaling=rope_scaling_of(t))
        if getattr(c, 'tie_word_embeddings', False):
            raise ValueError('llava with tied input / output embeddings has no native implementation (text-only llama / qwen2 checkpoints are built)')
        vision = clip_vision_cfg(v.hidden_size, v.intermediate_size, v.num_hidden_layers,
                                 v.num_attention_heads, v.image_size, v.patch_size, v.layer_norm_eps,
                                 v.num_channels)
        return llava_cfg(text, vision, c.image_token_id, c.vision_feature_layer,
                         getattr(c, 'pad_token_id', None))
    if mt == 'llama':
        rp = getattr(c, 'rope_parameters', None) or {}
        theta = rp.get('rope_theta', getattr(c, 'rope_theta', 10000.0))
        check_llama_family(c, 'llama')
        return llama_cfg(c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.num_attention_heads,
                         c.num_key_value_heads, c.vocab_size, c.rms_norm_eps, theta,
                         getattr(c, 'head_dim', None), c.max_position_embeddings, attention_bias=bool(getattr(c, 'attention_bias', False)),
                         rope_scaling=rope_scaling_of(c), tie_word_embeddings=getattr(c, 'tie_word_embeddings', False))
    if mt == 'qwen2':   # Llama block + q/k/v biases (align_anything/models/qwen2.py -> hf Qwen2ForCausalLM)
        rp = getattr(c, 'rope_parameters', None) or {}
        theta = rp.get('rope_theta', getattr(c, 'rope_theta', 1000000.0))
        if getattr(c, 'use_sliding_window', False):
            raise ValueError('qwen2 with sliding-window attention has no native implementation yet')
        check_llama_family(c, 'qwen2')
        return llama_cfg(c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.num_attention_heads,
                         c.num_key_value_heads, c.vocab_size, c.rms_norm_eps, theta, getattr(c, 'head_dim', None),
                         c.max_position_embeddings, attention_bias=True, rope_scaling=rope_scaling_of(c),
                         tie_word_embeddings=getattr(c, 'tie_word_embeddings', False))
    if mt == 'qwen2_vl':   # align_anything/models/qwen2_vl.py -> hf Qwen2VLForConditionalGeneration
        t, v = c.text_config, c.vision_config
        rp = getattr(t, 'rope_parameters', None) or getattr(t, 'rope_scaling', None) or {}
        theta = rp.get('rope_theta', getattr(t, 'rope_theta', 1000000.0))
        text = llama_cfg(t.hidden_size, t.intermediate_size, t.num_hidden_layers, t.num_attention_heads, t.num_key_value_heads,
                         t.vocab_size, t.rms_norm_eps, theta, getattr(t, 'head_dim', None), t.max_position_embeddings,
                         attention_bias=True)
        check_llama_family(t, 'qwen2_vl text_config')
        if getattr(v, 'hidden_act', 'quick_gelu') != 'quick_gelu' or getattr(t, 'use_sliding_window', False):
            raise ValueError('qwen2_vl: a vision activation other than quick_gelu / sliding-window attention has no native implementation')
        if getattr(c, 'tie_word_embeddings', False):
            raise ValueError('qwen2_vl with tied input / output embeddings (the 2B checkpoints) has no native implementation')
        if rope_scaling_of(t) is not None:
            raise ValueError('qwen2_vl: a scaled M-RoPE has no native implementation')
        vision = qwen2vl_vision_cfg(v.embed_dim, v.depth, v.num_heads, v.mlp_ratio, v.hidden_size, v.patch_size,
                                    v.temporal_patch_size, v.spatial_merge_size, v.in_channels)
        return qwen2vl_cfg(text, vision, c.image_token_id, rp['mrope_section'], getattr(c, 'pad_token_id', None))
    if mt == 'qwen2_audio':   # align_anything/models/qwen2_audio.py -> hf Qwen2AudioForConditionalGeneration
        t, a = c.text_config, c.audio_config
        rp = getattr(t, 'rope_parameters', None) or {}
        theta = rp.get('rope_theta', getattr(t, 'rope_theta', 10000.0))
        text = llama_cfg(t.hidden_size, t.intermediate_size, t.num_hidden_layers, t.num_attention_heads, t.num_key_value_heads,
                         t.vocab_size, t.rms_norm_eps, theta, getattr(t, 'head_dim', None), t.max_position_embeddings,
                         attention_bias=True)
        check_llama_family(t, 'qwen2_audio text_config')
        if getattr(a, 'activation_function', 'gelu') != 'gelu' or getattr(a, 'scale_embedding', False) or getattr(t, 'use_sliding_window', False):
            raise ValueError('qwen2_audio: an encoder activation other than gelu / scaled embeddings / sliding-window attention has no native implementation')
        if rope_scaling_of(t) is not None:
            raise ValueError('qwen2_audio: RoPE scaling has no native implementation')
        audio = qwen2audio_tower_cfg(a.d_model, a.encoder_layers, a.encoder_attention_heads, a.encoder_ffn_dim, a.num_mel_bins,
                                     a.max_source_positions)
        return qwen2audio_cfg(text, audio, c.audio_token_id, getattr(c, 'pad_token_id', None))
    if mt == 'qwen3_moe':   # align_anything/models/qwen3_moe.py -> hf Qwen3MoeForCausalLM
        rp = getattr(c, 'rope_parameters', None) or {}
        theta = rp.get('rope_theta', getattr(c, 'rope_theta', 1000000.0))
        if getattr(c, 'mlp_only_layers', None) or getattr(c, 'decoder_sparse_step', 1) != 1:
            raise ValueError('qwen3_moe with dense layers (mlp_only_layers / decoder_sparse_step) has no native implementation yet')
        if getattr(c, 'attention_bias', False):
            raise ValueError('qwen3_moe with attention_bias has no native implementation yet')
        if rope_scaling_of(c) is not None:
            raise ValueError('qwen3_moe: RoPE scaling has no native implementation')
        if getattr(c, 'hidden_act', 'silu') != 'silu' or getattr(c, 'use_sliding_window', False):
            raise ValueError('qwen3_moe: an activation other than silu / sliding-window attention has no native implementation')
        return qwen3moe_cfg(c.hidden_size, c.moe_intermediate_size, c.num_hidden_layers, c.num_attention_heads, c.num_key_value_heads,
                            c.vocab_size, c.num_experts, c.num_experts_per_tok, getattr(c, 'head_dim', None) or c.hidden_size // c.num_attention_heads,
                            c.norm_topk_prob, c.rms_norm_eps, theta, c.max_position_embeddings)
    if mt == 'opt':
        if not getattr(c, 'do_layer_norm_before', True) or getattr(c, 'word_embed_proj_dim', c.hidden_size) != c.hidden_size \
                or getattr(c, 'activation_function', 'relu') != 'relu' or not getattr(c, 'enable_bias', True) or not getattr(c, 'layer_norm_elementwise_affine', True):
            raise ValueError('opt: post-layer-norm blocks (opt-350m), a projected word embedding, a non-ReLU activation or bias-free layers have no native implementation')
        return opt_cfg(c.hidden_size, c.ffn_dim, c.num_hidden_layers, c.num_attention_heads, c.vocab_size,
                       c.max_position_embeddings)
    raise ValueError(f'model_type {mt!r} has no native MI355X implementation (llava, llama, qwen2, qwen2_vl, qwen2_audio, qwen3_moe, opt are built)')
