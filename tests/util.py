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
