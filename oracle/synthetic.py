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
