"""GPU tests written AFTER round 4's GPU budget was spent: never run on hardware yet.  They cover host-visible features whose device
arithmetic is shared with validated paths (tied embeddings = NativeOPT's shared gradient buffer on the Llama head; llama3 / linear RoPE
scaling = other values in the same cos / sin tables), so they are expected to pass -- but "expected" is not "measured": they are skipped
unless AA_GPU_UNVALIDATED=1, and the first GPU call of the next round runs them and removes the gate."""
import os

import pytest
import torch

from oracle import rl_math as orl
from tests.gpu_util import dev
from tests.util import rel_err

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get('AA_GPU_UNVALIDATED') != '1',
                                                  reason='not yet run on hardware (written after the round-4 GPU budget was spent); AA_GPU_UNVALIDATED=1 runs it')]


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_tied_embeddings_and_llama3_rope_dpo_step_vs_hf(dtype):
    """A Qwen2 / Llama-3.2-shaped decoder with `tie_word_embeddings` and Llama-3.1's RoPE scaling: DPO loss and gradients of the native path
    against transformers' own Qwen2ForCausalLM in fp32 with autograd (the arithmetic the reference runs, models/qwen2.py), positions past the
    original context included so that the scaled frequencies matter.  The tied matrix's gradient = head dW + embedding scatter."""
    import transformers as tf
    from align_anything_amd import configs
    from align_anything_amd.trainers.dpo import DPOTrainer
    rp = {'rope_type': 'llama3', 'rope_theta': 500000.0, 'factor': 8.0, 'low_freq_factor': 1.0, 'high_freq_factor': 4.0, 'original_max_position_embeddings': 32}
    hf_cfg = tf.Qwen2Config(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, vocab_size=320,
                            max_position_embeddings=128, rms_norm_eps=1e-6, tie_word_embeddings=True, rope_parameters=rp, attn_implementation='eager')
    torch.manual_seed(11)
    pol, ref = tf.Qwen2ForCausalLM(hf_cfg).float().eval(), tf.Qwen2ForCausalLM(hf_cfg).float().eval()
    with torch.no_grad():
        for p in list(pol.parameters()) + list(ref.parameters()):
            p.copy_(p.to(torch.bfloat16).float())                 # bf16-representable: both dtypes load identical numbers
    cfg = configs.from_hf_config(hf_cfg)
    assert cfg['tie_word_embeddings'] and cfg['rope_scaling']['type'] == 'llama3'
    tr = DPOTrainer({'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-3, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant', 'compute_dtype': dtype},
                     'model_cfgs': {'pad_token_id': 0}}, {'gradient_clipping': 1.0}, model_cfg=cfg,
                    policy_state={k: v for k, v in pol.state_dict().items()}, reference_state={k: v for k, v in ref.state_dict().items()}, device='cuda:0')
    assert tr.policy.tied
    g = torch.Generator().manual_seed(9)
    N, Tn = 4, 96
    ids = torch.randint(3, 320, (N, Tn), generator=g)
    mask = torch.ones(N, Tn, dtype=torch.long)
    for n, lp in enumerate((0, 7, 3, 0)):
        ids[n, :lp] = 0
        mask[n, :lp] = 0
    lens = [9, 12, 5, 16]
    ld = tr.loss({'input_ids': ids.to(dev()), 'attention_mask': mask.to(dev()), 'meta_info': {'response_lens': lens}})
    lp = orl.compute_log_probs(pol(input_ids=ids, attention_mask=mask).logits, ids, lens, 0)
    with torch.no_grad():
        rlp = orl.compute_log_probs(ref(input_ids=ids, attention_mask=mask).logits, ids, lens, 0)
    o = orl.dpo_loss(lp, rlp, 0.1)
    tight = dtype == 'fp32'
    assert abs(float(ld['loss']) - float(o['loss'])) < (2e-5 if tight else 1e-2), (float(ld['loss']), float(o['loss']))
    tr.model.backward(ld['loss'])
    torch.cuda.synchronize()
    o['loss'].backward()
    want = dict(pol.named_parameters())
    for n in ('model.embed_tokens.weight', 'model.layers.0.self_attn.q_proj.weight', 'model.layers.1.self_attn.k_proj.bias', 'model.layers.1.mlp.down_proj.weight'):
        got = tr.policy.store.grad_view(n).float().cpu().reshape(want[n].grad.shape)
        assert rel_err(got, want[n].grad) < (2e-4 if tight else 8e-2), (n, rel_err(got, want[n].grad))
