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
