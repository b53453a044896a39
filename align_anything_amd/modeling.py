"""Native MI355X transformer modules with EXPLICIT forward and backward (no autograd, no HF modules).

These replace, on the hot path, what `model(**batch).logits` executes in the reference
(align_anything/trainers/text_to_text/dpo.py:128): AccustomedLlavaModel / AccustomedOPTModel
(align_anything/models/llava.py:29, models/opt.py:28), i.e. HF LlavaForConditionalGeneration /
OPTForCausalLM.  Every arithmetic op is a call into libaa_hip.so (ops.py); python only sequences
launches and owns buffers.  Parameters carry the HF state-dict names, so checkpoints written by
`save_pretrained` load directly and what we save loads back through `AnyModel.from_pretrained`.

Activations are token-major [Mp, h] with Mp = N*T rounded up to 64 (pad rows are zero and stay finite), so
the backward dW GEMMs (contraction over tokens) meet the K % 64 rule of the MFMA kernel.
Memory: nothing is recomputed -- 288 GB HBM holds all saved activations of a 7B model at seq 2048
(DESIGN.md §memory), so `gradient_checkpointing_enable()` is accepted and ignored.
"""
from __future__ import annotations

import math

import torch

from . import ops
from .params import ParamStore

bf16 = torch.bfloat16


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def rope_tables(max_pos: int, hd: int, theta: float, device, dtype=bf16):
    """hf:models/llama/modeling_llama.py:113-127: inv_freq / freqs in fp32, cos & sin cast to the activation dtype."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.int64).to(torch.float32) / hd))
    freqs = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv_freq[None, :]
    return freqs.cos().to(dtype).to(device).contiguous(), freqs.sin().to(dtype).to(device).contiguous()


class Linear:
    """y = x W^T (+ b).  dX via the NN layout GEMM, dW via the TN layout GEMM (no transposes)."""

    def __init__(self, store: ParamStore, wname: str, bname: str | None = None):
        self.store, self.wname, self.bname = store, wname, bname

    @property
    def w(self):
        return self.store.p[self.wname] if self.wname in self.store.p else self.store.view(self.wname)

    @property
    def b(self):
        return None if self.bname is None else self.store.p[self.bname]

    def fwd(self, x, residual=None, act=0, out=None):
        return ops.gemm(x, self.w, out=out, bias=self.b, residual=residual, act=act)

    def dx(self, dy, out=None):
        return ops.gemm(dy, self.w, out=out, b_n=True)

    def dw(self, dy, x):
        gw = self.store.g.get(self.wname)
        if gw is not None:
            ops.gemm(dy, x, out=gw, a_t=True, b_n=True, accumulate=(gw.dtype == torch.float32) or self.store.accumulate)
        if self.bname is not None and self.bname in self.store.g:
            ops.colsum_(dy, self.store.g[self.bname])


# ====================================================================== Llama decoder stack
class LlamaStack:
    """hf:models/llama/modeling_llama.py:295-325 x num_layers; q/k/v and gate/up stored fused."""

    def __init__(self, cfg: dict, store: ParamStore, prefix: str, trainable: bool):
        self.cfg, self.store, self.prefix, self.trainable = cfg, store, prefix, trainable
        h, F = cfg['hidden_size'], cfg['intermediate_size']
        H, Hkv, hd = cfg['num_heads'], cfg['num_kv_heads'], cfg['head_dim']
        if hd not in (64, 128):
            raise NotImplementedError(f'head_dim {hd}: attention kernels are built for 64 and 128')
        self.layers = []
        for i in range(cfg['num_layers']):
            p = f'{prefix}layers.{i}.'
            L = {}
            L['ln1'] = store.add(p + 'input_layernorm.weight', (h,), trainable)
            blk = store.add_fused(p + 'self_attn.qkv_fused', [(p + 'self_attn.q_proj.weight', H * hd, h),
                                                              (p + 'self_attn.k_proj.weight', Hkv * hd, h),
                                                              (p + 'self_attn.v_proj.weight', Hkv * hd, h)], trainable)
            bq = None
            if cfg.get('attention_bias'):   # Qwen2-style q/k/v biases (hf:models/qwen2/modeling_qwen2.py), fused [q|k|v]
                bq = store.add(p + 'self_attn.qkv_fused.bias', ((H + 2 * Hkv) * hd,), trainable)
                del store.alias[bq]
                off = 0
                for nm, rows in (('q', H * hd), ('k', Hkv * hd), ('v', Hkv * hd)):
                    store.alias[p + f'self_attn.{nm}_proj.bias'] = (bq, off, (rows,))
                    off += rows
            L['qkv'] = Linear(store, blk, bq)
            L['o'] = Linear(store, store.add(p + 'self_attn.o_proj.weight', (h, H * hd), trainable))
            L['ln2'] = store.add(p + 'post_attention_layernorm.weight', (h,), trainable)
            blk = store.add_fused(p + 'mlp.gate_up_fused', [(p + 'mlp.gate_proj.weight', F, h),
                                                            (p + 'mlp.up_proj.weight', F, h)], trainable)
            L['gu'] = Linear(store, blk)
            L['down'] = Linear(store, store.add(p + 'mlp.down_proj.weight', (h, F), trainable))
            self.layers.append(L)
        self.norm = store.add(prefix + 'norm.weight', (h,), trainable)
        self.cos = self.sin = None
        self.saved = []

    def _tables(self, T):
        if self.cos is None or self.cos.shape[0] < T:
            n = max(T, self.cfg.get('max_position_embeddings', 0) or T)
            self.cos, self.sin = rope_tables(n, self.cfg['head_dim'], self.cfg['rope_theta'], self.store.device, self.store.dtype)

    def forward(self, x, N, T, start, pos, save, kv_sink=None):
        c, P = self.cfg, self.store.p
        H, Hkv, hd, eps = c['num_heads'], c['num_kv_heads'], c['head_dim'], c['rms_eps']
        self._tables(T)
        self.saved = []
        qw, kw = H * hd, Hkv * hd
        for li, L in enumerate(self.layers):
            n1, rstd1 = ops.rmsnorm_fwd(x, P[L['ln1']], eps)
            qkv = L['qkv'].fwd(n1)
            ops.rope_(qkv, 0, H + Hkv, hd, pos, self.cos, self.sin)
            if kv_sink is not None:
                kv_sink(li, qkv[:N * T, qw:])  # post-RoPE keys | values of this layer -> KV cache (prefill)
            attn, lse = ops.attn_fwd(qkv[:, :qw], qkv[:, qw:qw + kw], qkv[:, qw + kw:], N, T, H, Hkv, hd, True,
                                     hd ** -0.5, start, out=self._attn_out(x, N * T, H * hd))
            x_mid = L['o'].fwd(attn, residual=x)
            n2, rstd2 = ops.rmsnorm_fwd(x_mid, P[L['ln2']], eps)
            gu = L['gu'].fwd(n2)
            act = ops.swiglu_fwd(gu)
            x_out = L['down'].fwd(act, residual=x_mid)
            if save:
                self.saved.append((x, rstd1, n1, qkv, attn, lse, x_mid, rstd2, n2, gu, act))
            x = x_out
        return x

    def kv_width(self):
        return 2 * self.cfg['num_kv_heads'] * self.cfg['head_dim']

    def decode_step(self, x, cache, t, Tmax, pos, start, length):
        """One new token per sequence (x [N, h]) against the KV cache (csrc/decode.hip): every GEMM streams its
        weight once through the skinny kernel.  cache[l]: [N*Tmax, 2*kw] (keys | values), slot t is written here."""
        c, P = self.cfg, self.store.p
        H, Hkv, hd, eps = c['num_heads'], c['num_kv_heads'], c['head_dim'], c['rms_eps']
        qw, kw = H * hd, Hkv * hd
        N = x.shape[0]
        self._tables(Tmax)
        rows = torch.arange(N, device=x.device)
        for li, L in enumerate(self.layers):
            n1, _ = ops.rmsnorm_fwd(x, P[L['ln1']], eps)
            qkv = ops.linear_small(n1, L['qkv'].w, bias=L['qkv'].b)
            ops.rope_(qkv, 0, H + Hkv, hd, pos, self.cos, self.sin)
            cl = cache[li]
            cl.view(N, Tmax, 2 * kw).index_put_((rows, t), qkv[:, qw:])   # t: device int64 [N] (graph-capturable)
            attn = ops.attn_decode(qkv[:, :qw], cl, cl[:, kw:], Tmax, start, length, N, H, Hkv, hd, hd ** -0.5)
            x_mid = ops.linear_small(attn, L['o'].w, residual=x)
            n2, _ = ops.rmsnorm_fwd(x_mid, P[L['ln2']], eps)
            gu = ops.linear_small(n2, L['gu'].w)
            act = ops.swiglu_fwd(gu)
            x = ops.linear_small(act, L['down'].w, residual=x_mid)
        return x

    @staticmethod
    def _attn_out(x, real_rows, width):
        # pad rows (Mp > N*T) are never written by the attention kernel: they must read as zeros
        return None if x.shape[0] == real_rows else torch.zeros((x.shape[0], width), dtype=x.dtype, device=x.device)

    def backward(self, dres, N, T, start, pos, on_layer_done=None):
        """dres: gradient of the residual stream after the last layer [Mp, h]; updated in place and returned
        as the gradient w.r.t. the stack input.  Weight gradients go to store.g."""
        c, P, G = self.cfg, self.store.p, self.store.g
        H, Hkv, hd = c['num_heads'], c['num_kv_heads'], c['head_dim']
        qw, kw = H * hd, Hkv * hd
        tr = self.trainable
        for L, sv in zip(reversed(self.layers), reversed(self.saved)):
            x, rstd1, n1, qkv, attn, lse, x_mid, rstd2, n2, gu, act = sv
            sv = None
            # ---- MLP
            d_act = L['down'].dx(dres)
            if tr:
                L['down'].dw(dres, act)
            d_gu = ops.swiglu_bwd(gu, d_act)
            d_n2 = L['gu'].dx(d_gu)
            if tr:
                L['gu'].dw(d_gu, n2)
            ops.rmsnorm_bwd(d_n2, x_mid, P[L['ln2']], rstd2, G.get(L['ln2']) if tr else None, dx=dres, add_to_dx=True)
            # ---- attention
            d_attn = L['o'].dx(dres)
            if tr:
                L['o'].dw(dres, attn)
            d_qkv = torch.zeros_like(qkv) if qkv.shape[0] != N * T else torch.empty_like(qkv)
            ops.attn_bwd(qkv[:, :qw], qkv[:, qw:qw + kw], qkv[:, qw + kw:], attn, d_attn, lse,
                         d_qkv[:, :qw], d_qkv[:, qw:qw + kw], d_qkv[:, qw + kw:], N, T, H, Hkv, hd, True,
                         hd ** -0.5, start)
            ops.rope_(d_qkv, 0, H + Hkv, hd, pos, self.cos, self.sin, inverse=True)
            d_n1 = L['qkv'].dx(d_qkv)
            if tr:
                L['qkv'].dw(d_qkv, n1)
            ops.rmsnorm_bwd(d_n1, x, P[L['ln1']], rstd1, G.get(L['ln1']) if tr else None, dx=dres, add_to_dx=True)
            if on_layer_done is not None:
                on_layer_done(L)
        self.saved = []
        return dres


# ====================================================================== CLIP vision tower (frozen: forward only)
class ClipVisionTower:
    """hf:models/clip/modeling_clip.py:138-218, :338-351, :605-651.  Runs layers 0..L+feature_layer only (the
    reference computes the last block(s) and post_layernorm and discards them, modeling_llava.py:154-166)."""

    def __init__(self, vcfg: dict, store: ParamStore, prefix: str, feature_layer: int = -2):
        self.cfg, self.store, self.prefix = vcfg, store, prefix
        h, F = vcfg['hidden_size'], vcfg['intermediate_size']
        self.K = vcfg['num_channels'] * vcfg['patch_size'] ** 2
        self.Kp = _pad64(self.K)
        self.G2 = (vcfg['image_size'] // vcfg['patch_size']) ** 2
        if h // vcfg['num_heads'] not in (64, 128):
            raise NotImplementedError('CLIP head_dim must be 64 or 128 for the native attention kernel')
        e = prefix + 'embeddings.'
        self.patch_w = store.add(e + 'patch_embedding.weight', (h, self.Kp), False)
        self.cls = store.add(e + 'class_embedding', (h,), False)
        self.pos = store.add(e + 'position_embedding.weight', (self.G2 + 1, h), False)
        self.pre_w = store.add(prefix + 'pre_layrnorm.weight', (h,), False)
        self.pre_b = store.add(prefix + 'pre_layrnorm.bias', (h,), False)
        self.layers = []
        for i in range(vcfg['num_layers']):
            p = f'{prefix}encoder.layers.{i}.'
            L = {}
            L['ln1w'] = store.add(p + 'layer_norm1.weight', (h,), False)
            L['ln1b'] = store.add(p + 'layer_norm1.bias', (h,), False)
            wq = store.add_fused(p + 'self_attn.qkv_fused.weight', [(p + f'self_attn.{n}_proj.weight', h, h) for n in 'qkv'], False)
            # biases are 1-D: fuse by registering one [3h] block with three aliases
            bq = store.add(p + 'self_attn.qkv_fused.bias', (3 * h,), False)
            del store.alias[bq]
            for j, n in enumerate('qkv'):
                store.alias[p + f'self_attn.{n}_proj.bias'] = (bq, j * h, (h,))
            L['qkv'] = Linear(store, wq, bq)
            L['out'] = Linear(store, store.add(p + 'self_attn.out_proj.weight', (h, h), False),
                              store.add(p + 'self_attn.out_proj.bias', (h,), False))
            L['ln2w'] = store.add(p + 'layer_norm2.weight', (h,), False)
            L['ln2b'] = store.add(p + 'layer_norm2.bias', (h,), False)
            L['fc1'] = Linear(store, store.add(p + 'mlp.fc1.weight', (F, h), False), store.add(p + 'mlp.fc1.bias', (F,), False))
            L['fc2'] = Linear(store, store.add(p + 'mlp.fc2.weight', (h, F), False), store.add(p + 'mlp.fc2.bias', (h,), False))
            self.layers.append(L)
        self.post_w = store.add(prefix + 'post_layernorm.weight', (h,), False)
        self.post_b = store.add(prefix + 'post_layernorm.bias', (h,), False)
        nl = vcfg['num_layers']
        self.run_layers = nl + 1 + feature_layer if feature_layer < 0 else feature_layer  # hidden_states index
        self._drop_cls_idx = {}

    def forward(self, pixel_values):
        """pixel_values [n, 3, S, S] (fp32 or bf16) -> patch features [n * G2, h] (CLS dropped)."""
        c, P = self.cfg, self.store.p
        n = pixel_values.shape[0]
        h, H = c['hidden_size'], c['num_heads']
        hd, eps, T = h // H, c['ln_eps'], self.G2 + 1
        col = ops.patch_im2col(pixel_values.contiguous(), c['patch_size'], self.Kp, self.store.dtype)
        pe = ops.gemm(col, P[self.patch_w])
        x = ops.clip_embed(pe, P[self.cls], P[self.pos], n, self.G2)
        x, _, _ = ops.layernorm_fwd(x, P[self.pre_w], P[self.pre_b], eps, want_stats=False)
        for L in self.layers[:self.run_layers]:
            y, _, _ = ops.layernorm_fwd(x, P[L['ln1w']], P[L['ln1b']], eps, want_stats=False)
            qkv = L['qkv'].fwd(y)
            a, _ = ops.attn_fwd(qkv[:, :h], qkv[:, h:2 * h], qkv[:, 2 * h:], n, T, H, H, hd, False, hd ** -0.5)
            x = L['out'].fwd(a, residual=x)
            y, _, _ = ops.layernorm_fwd(x, P[L['ln2w']], P[L['ln2b']], eps, want_stats=False)
            y = L['fc1'].fwd(y, act=ops.ACT_QUICK_GELU)
            x = L['fc2'].fwd(y, residual=x)
        idx = self._drop_cls_idx.get(n)
        if idx is None:
            idx = (torch.arange(n * T).view(n, T)[:, 1:]).reshape(-1).to(self.store.device)
            self._drop_cls_idx[n] = idx
        return ops.embed_fwd(idx, x)  # row gather: drop the CLS token of every image


# ====================================================================== LM head + log-prob (shared)
class LMHead:
    """final norm -> lm_head -> log_softmax/gather, evaluated ONLY on the response-window rows
    (the reference computes logits for all T positions and slices, dpo.py:128-136)."""

    def __init__(self, store, norm_kind, norm_w, norm_b, lm_w, eps, trainable):
        self.store, self.kind, self.norm_w, self.norm_b, self.lm_w = store, norm_kind, norm_w, norm_b, lm_w
        self.eps, self.trainable = eps, trainable
        self.saved = None

    def _w(self):
        return self.store.p[self.lm_w] if self.lm_w in self.store.p else self.store.view(self.lm_w)

    def forward(self, x_last, row_idx, labels, save, round_bf16=False):
        P = self.store.p
        sel = ops.embed_fwd(row_idx, x_last)
        if self.kind == 'rms':
            n, rstd = ops.rmsnorm_fwd(sel, P[self.norm_w], self.eps)
            mean = None
        else:
            n, mean, rstd = ops.layernorm_fwd(sel, P[self.norm_w], P[self.norm_b], self.eps)
        logits = ops.gemm(n, self._w())
        logp, lse = ops.logprob_gather_fwd(logits, labels, round_bf16)
        self.saved = (sel, mean, rstd, n, logits, lse, labels) if save else None
        return logp

    def logits_all(self, x_last):
        """Full-vocabulary logits for every row (parity tests / generation); not used by the DPO step."""
        P = self.store.p
        if self.kind == 'rms':
            n, _ = ops.rmsnorm_fwd(x_last, P[self.norm_w], self.eps)
        else:
            n, _, _ = ops.layernorm_fwd(x_last, P[self.norm_w], P[self.norm_b], self.eps, want_stats=False)
        return ops.gemm(n, self._w())

    def logits_rows(self, x_rows):
        """Logits of a handful of rows (decode): norm + skinny lm_head."""
        P = self.store.p
        if self.kind == 'rms':
            n, _ = ops.rmsnorm_fwd(x_rows, P[self.norm_w], self.eps)
        else:
            n, _, _ = ops.layernorm_fwd(x_rows, P[self.norm_w], P[self.norm_b], self.eps, want_stats=False)
        return ops.linear_small(n, self._w())

    def hidden_all(self, x_last):
        P = self.store.p
        if self.kind == 'rms':
            return ops.rmsnorm_fwd(x_last, P[self.norm_w], self.eps)[0]
        return ops.layernorm_fwd(x_last, P[self.norm_w], P[self.norm_b], self.eps, want_stats=False)[0]

    def backward(self, dlogp, inv_map, zero_row):
        """dlogp f32[rows_pad] -> gradient of the residual stream [Mp, h] (rows outside the windows are 0)."""
        sel, mean, rstd, n, logits, lse, labels = self.saved
        P, G = self.store.p, self.store.g
        ops.logprob_gather_bwd(logits, labels, lse, dlogp, out=logits)  # in place: logits -> dlogits
        d_n = ops.gemm(logits, self._w(), b_n=True)
        if self.trainable:
            gw = G.get(self.lm_w)
            if gw is None:
                gw = self.store.grad_view(self.lm_w)
            ops.gemm(logits, n, out=gw, a_t=True, b_n=True, accumulate=(gw.dtype == torch.float32) or self.store.accumulate)
        tr = self.trainable
        if self.kind == 'rms':
            d_sel = ops.rmsnorm_bwd(d_n, sel, P[self.norm_w], rstd, G.get(self.norm_w) if tr else None)
        else:
            d_sel = ops.layernorm_bwd(d_n, sel, P[self.norm_w], mean, rstd, G.get(self.norm_w) if tr else None,
                                      G.get(self.norm_b) if tr else None)
        self.saved = None
        zeros_ids = torch.zeros(inv_map.shape[0], dtype=torch.int64, device=inv_map.device)
        return ops.embed_fwd(zeros_ids, zero_row, slot=inv_map, feat=d_sel)


class ScoreHead:
    """final norm -> score_head Linear(h -> 1, no bias) on the selected rows: the reward / critic models of the
    reference (align_anything/models/opt.py:45-97, models/llava.py:47-76; `scores` fp32)."""

    def __init__(self, store, norm_kind, norm_w, norm_b, score_w, eps, trainable):
        self.store, self.kind, self.norm_w, self.norm_b, self.score_w = store, norm_kind, norm_w, norm_b, score_w
        self.eps, self.trainable = eps, trainable
        self.saved = None

    def _norm(self, x, stats):
        P = self.store.p
        if self.kind == 'rms':
            n, rstd = ops.rmsnorm_fwd(x, P[self.norm_w], self.eps)
            return n, None, rstd
        n, mean, rstd = ops.layernorm_fwd(x, P[self.norm_w], P[self.norm_b], self.eps, want_stats=stats)
        return n, mean, rstd

    def forward(self, x_last, row_idx, labels=None, save=False, round_bf16=False):
        sel = ops.embed_fwd(row_idx, x_last)
        n, mean, rstd = self._norm(sel, True)
        scores = ops.rowdot_fwd(n, self.store.p[self.score_w].view(-1))
        self.saved = (sel, mean, rstd, n) if save else None
        return scores

    def hidden_all(self, x_last):
        return self._norm(x_last, False)[0]

    def scores_all(self, x_last):
        return ops.rowdot_fwd(self.hidden_all(x_last), self.store.p[self.score_w].view(-1))

    def backward(self, dscores, inv_map, zero_row):
        sel, mean, rstd, n = self.saved
        P, G = self.store.p, self.store.g
        tr = self.trainable
        gw = G.get(self.score_w) if tr else None
        d_n = ops.rowdot_bwd(dscores, n, P[self.score_w].view(-1), gw.view(-1) if gw is not None else None)
        if self.kind == 'rms':
            d_sel = ops.rmsnorm_bwd(d_n, sel, P[self.norm_w], rstd, G.get(self.norm_w) if tr else None)
        else:
            d_sel = ops.layernorm_bwd(d_n, sel, P[self.norm_w], mean, rstd, G.get(self.norm_w) if tr else None,
                                      G.get(self.norm_b) if tr else None)
        self.saved = None
        zeros_ids = torch.zeros(inv_map.shape[0], dtype=torch.int64, device=inv_map.device)
        return ops.embed_fwd(zeros_ids, zero_row, slot=inv_map, feat=d_sel)


# ====================================================================== model base
class NativeCausalLM:
    """Common driver: batch contract of the reference collators in, per-token response log-probs out."""

    kind = 'base'

    def __init__(self, cfg: dict, device, trainable: bool = True, dtype=bf16):
        self.cfg, self.device, self.trainable = cfg, torch.device(device), trainable
        self.dtype = dtype          # bf16 = production; float32 = parity mode (include/aa_hip_f32.h kernels)
        self.store = ParamStore(device, dtype)
        self.training = trainable
        self._ctx = None
        self._zero_row = None

    # -- HF-shaped conveniences used by the trainers
    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def parameters_count(self):
        return self.store.num_params()

    def state_dict(self):
        return self.store.state_dict(self._unpad())

    def load_state_dict(self, sd, strict=True):
        return self.store.load_state_dict(sd, strict, self._padcols())

    def _unpad(self):
        return None

    def _padcols(self):
        return None

    def finalize(self):
        self.store.allocate()
        self._zero_row = torch.zeros((1, self.hidden_size), dtype=self.dtype, device=self.device)

    def init_training(self):
        self.store.init_training()

    # -- geometry helpers
    def _token_geometry(self, input_ids, attention_mask, position_ids=None):
        N, T = input_ids.shape
        Mp = _pad64(N * T)
        if attention_mask is not None:
            start = attention_mask.to(torch.int32).argmax(dim=1).to(torch.int32)  # first attended key
        else:
            start = None
        if position_ids is not None:  # HF generate(): positions from the mask (SURVEY.md §8 a' quirk)
            pos = position_ids.to(torch.int32).reshape(-1)
        else:
            pos = torch.arange(T, dtype=torch.int32, device=self.device).repeat(N)
        if Mp != N * T:
            pos = torch.cat([pos, torch.zeros(Mp - N * T, dtype=torch.int32, device=self.device)])
        return N, T, Mp, start, pos

    def forward_stream(self, input_ids, attention_mask=None, pixel_values=None, save=False, image_features=None,
                       position_ids=None, kv_sink=None):
        raise NotImplementedError

    def backward_stream(self, dres, on_layer_done=None):
        raise NotImplementedError

    # -- the DPO/PPO entry points
    def response_logprobs(self, input_ids, attention_mask, window, pixel_values=None, save=False,
                          image_features=None, round_bf16=False):
        """window: dict(row_idx int64[rows_pad], labels int64[rows_pad], inv_map int32[Mp]) built by
        trainers.common.build_window.  Returns flat fp32 log-probs [rows_pad] (pad rows meaningless)."""
        x = self.forward_stream(input_ids, attention_mask, pixel_values, save, image_features)
        logp = self.head.forward(x, window['row_idx'], window['labels'], save, round_bf16)
        if save:
            self._ctx['window'] = window
        return logp

    def response_scores(self, input_ids, attention_mask, window, pixel_values=None, save=False, image_features=None):
        """Score-head models: fp32 scores on the window rows (critic values / reward scores)."""
        x = self.forward_stream(input_ids, attention_mask, pixel_values, save, image_features)
        sc = self.head.forward(x, window['row_idx'], None, save)
        if save:
            self._ctx['window'] = window
        return sc

    def scores(self, input_ids, attention_mask=None, pixel_values=None):
        """ScoreModelOutput.scores [N, T] (fp32) for every position."""
        N, T = input_ids.shape
        x = self.forward_stream(input_ids, attention_mask, pixel_values, save=False)
        return self.head.scores_all(x)[:N * T].view(N, T)

    def backward_from_dlogp(self, dlogp, on_layer_done=None):
        dres = self.head.backward(dlogp, self._ctx['window']['inv_map'], self._zero_row)
        self.backward_stream(dres, on_layer_done)
        self._ctx = None

    def logits(self, input_ids, attention_mask=None, pixel_values=None):
        """All-position logits [N, T, V] (what HF returns) -- parity tests and PPO/generation callers."""
        N, T = input_ids.shape
        x = self.forward_stream(input_ids, attention_mask, pixel_values, save=False)
        return self.head.logits_all(x)[:N * T].view(N, T, -1)

    def final_hidden(self, input_ids, attention_mask=None, pixel_values=None):
        N, T = input_ids.shape
        x = self.forward_stream(input_ids, attention_mask, pixel_values, save=False)
        return self.head.hidden_all(x)[:N * T].view(N, T, -1)


# ====================================================================== LLaVA
class NativeLlava(NativeCausalLM):
    """hf:models/llava/modeling_llava.py:301-371 LlavaForConditionalGeneration, natively."""

    kind = 'llava'

    def __init__(self, cfg, device, trainable=True, freeze_mm_proj=False, freeze_language_model=False,
                 freeze_vision_tower=True, head='lm', dtype=bf16):
        super().__init__(cfg, device, trainable, dtype)
        self.head_kind = head
        if not freeze_vision_tower and trainable:
            raise NotImplementedError('training the CLIP vision tower is not built (the reference default freezes '
                                      'it, configs/train/text_image_to_text/dpo.yaml:60)')
        t = cfg['text']
        self.hidden_size = t['hidden_size']
        self.train_lm = trainable and not freeze_language_model
        self.train_proj = trainable and not freeze_mm_proj
        st = self.store
        self.vision = ClipVisionTower(cfg['vision'], st, 'model.vision_tower.', cfg.get('vision_feature_layer', -2))
        vh = cfg['vision']['hidden_size']
        self.proj1 = Linear(st, st.add('model.multi_modal_projector.linear_1.weight', (t['hidden_size'], vh), self.train_proj),
                            st.add('model.multi_modal_projector.linear_1.bias', (t['hidden_size'],), self.train_proj))
        self.proj2 = Linear(st, st.add('model.multi_modal_projector.linear_2.weight', (t['hidden_size'], t['hidden_size']), self.train_proj),
                            st.add('model.multi_modal_projector.linear_2.bias', (t['hidden_size'],), self.train_proj))
        self.embed = st.add('model.language_model.embed_tokens.weight', (t['vocab_size'], t['hidden_size']), self.train_lm, f32_grad=True)
        self.stack = LlamaStack(t, st, 'model.language_model.', self.train_lm)
        if head == 'lm':
            lm = st.add('lm_head.weight', (t['vocab_size'], t['hidden_size']), self.train_lm)
            self.head = LMHead(st, 'rms', self.stack.norm, None, lm, t['rms_eps'], self.train_lm)
        else:  # reward / critic: AccustomedLlavaRewardModel (models/llava.py:35-76)
            sw = st.add('score_head.weight', (1, t['hidden_size']), trainable, f32_grad=True)
            self.head = ScoreHead(st, 'rms', self.stack.norm, None, sw, t['rms_eps'], trainable)
        self.finalize()

    def _padcols(self):
        return {'model.vision_tower.embeddings.patch_embedding.weight': True}

    def _unpad(self):
        v = self.cfg['vision']
        return {'model.vision_tower.embeddings.patch_embedding.weight':
                (v['hidden_size'], v['num_channels'], v['patch_size'], v['patch_size'])}

    def vision_features(self, pixel_values):
        """Frozen tower: callers may run it ONCE per unique image and share the result between the
        chosen/rejected rows and between policy and reference (the reference runs it 4x per pair)."""
        return self.vision.forward(pixel_values)

    def forward_stream(self, input_ids, attention_mask=None, pixel_values=None, save=False, image_features=None,
                       position_ids=None, kv_sink=None):
        N, T, Mp, start, pos = self._token_geometry(input_ids, attention_mask, position_ids)
        P = self.store.p
        ids = input_ids.reshape(-1)
        if Mp != N * T:
            ids = torch.cat([ids, torch.zeros(Mp - N * T, dtype=ids.dtype, device=ids.device)])
        slot = feat = f1 = a1 = vfeat = None
        if pixel_values is not None or image_features is not None:
            vfeat = image_features if image_features is not None else self.vision.forward(pixel_values)
            n_feat = vfeat.shape[0]
            if n_feat % 64:  # rows are the contraction dim of the projector dW GEMMs (K % 64); pad rows are zero
                vfeat = torch.cat([vfeat, torch.zeros((_pad64(n_feat) - n_feat, vfeat.shape[1]), dtype=vfeat.dtype, device=vfeat.device)])
            f1 = self.proj1.fwd(vfeat)
            a1 = ops.act_fwd(f1, ops.ACT_GELU)
            feat = self.proj2.fwd(a1)
            slot, count = ops.image_slot_index(ids, self.cfg['image_token_id'])
            self._last_image_token_count = count  # device scalar; checked lazily by validate_batch()
            self._last_feature_rows = n_feat
        x = ops.embed_fwd(ids, P[self.embed], slot, feat)
        if save:
            self._ctx = dict(ids=ids, slot=slot, f1=f1, a1=a1, vfeat=vfeat, N=N, T=T, start=start, pos=pos)
        return self.stack.forward(x, N, T, start, pos, save, kv_sink)

    def embed_tokens(self, ids, pos=None):
        return ops.embed_fwd(ids, self.store.p[self.embed])

    def validate_batch(self):
        """hf:models/llava/modeling_llava.py:191-213 raises when #image tokens != #features; this is the same
        check, done on demand because it needs a device->host sync."""
        c = int(self._last_image_token_count.item())
        if c != self._last_feature_rows:
            raise ValueError(f'Image features and image tokens do not match: tokens: {c}, features {self._last_feature_rows}')

    def backward_stream(self, dres, on_layer_done=None):
        cx = self._ctx
        dx = self.stack.backward(dres, cx['N'], cx['T'], cx['start'], cx['pos'], on_layer_done)
        G = self.store.g
        want_feat = cx['slot'] is not None and self.train_proj
        dfeat = torch.zeros((cx['vfeat'].shape[0], self.hidden_size), dtype=self.dtype, device=self.device) if want_feat else None
        if self.train_lm or want_feat:
            ops.embed_bwd(cx['ids'], dx, self.cfg['text']['vocab_size'], slot=cx['slot'],
                          dE=G.get(self.embed) if self.train_lm else None, dfeat=dfeat)
        if want_feat:
            self.proj2.dw(dfeat, cx['a1'])
            d_a1 = self.proj2.dx(dfeat)
            d_f1 = ops.act_bwd(cx['f1'], d_a1, ops.ACT_GELU)
            self.proj1.dw(d_f1, cx['vfeat'])


# ====================================================================== Llama (text only)
class NativeLlama(NativeCausalLM):
    """hf:models/llama/modeling_llama.py LlamaForCausalLM (MHA or GQA, head_dim 64/128): the text-to-text trainers'
    backbone for Llama-family checkpoints (align_anything/models/llama.py)."""

    kind = 'llama'

    def __init__(self, cfg, device, trainable=True, head='lm', dtype=bf16):
        super().__init__(cfg, device, trainable, dtype)
        self.head_kind = head
        self.hidden_size = cfg['hidden_size']
        st = self.store
        self.embed = st.add('model.embed_tokens.weight', (cfg['vocab_size'], cfg['hidden_size']), trainable, f32_grad=True)
        self.stack = LlamaStack(cfg, st, 'model.', trainable)
        if head == 'lm':
            lm = st.add('lm_head.weight', (cfg['vocab_size'], cfg['hidden_size']), trainable)
            self.head = LMHead(st, 'rms', self.stack.norm, None, lm, cfg['rms_eps'], trainable)
        else:
            sw = st.add('score_head.weight', (1, cfg['hidden_size']), trainable, f32_grad=True)
            self.head = ScoreHead(st, 'rms', self.stack.norm, None, sw, cfg['rms_eps'], trainable)
        self.finalize()

    def forward_stream(self, input_ids, attention_mask=None, pixel_values=None, save=False, image_features=None,
                       position_ids=None, kv_sink=None):
        N, T, Mp, start, pos = self._token_geometry(input_ids, attention_mask, position_ids)
        ids = input_ids.reshape(-1)
        if Mp != N * T:
            ids = torch.cat([ids, torch.zeros(Mp - N * T, dtype=ids.dtype, device=ids.device)])
        x = ops.embed_fwd(ids, self.store.p[self.embed])
        if save:
            self._ctx = dict(ids=ids, N=N, T=T, start=start, pos=pos)
        return self.stack.forward(x, N, T, start, pos, save, kv_sink)

    def embed_tokens(self, ids, pos=None):
        return ops.embed_fwd(ids, self.store.p[self.embed])

    def backward_stream(self, dres, on_layer_done=None):
        cx = self._ctx
        dx = self.stack.backward(dres, cx['N'], cx['T'], cx['start'], cx['pos'], on_layer_done)
        if self.trainable:
            ops.embed_bwd(cx['ids'], dx, self.cfg['vocab_size'], dE=self.store.g[self.embed])


# ====================================================================== OPT
class OPTStack:
    """hf:models/opt/modeling_opt.py:191-251 pre-LN decoder layer (biased projections, ReLU MLP)."""

    def __init__(self, cfg, store, prefix, trainable):
        self.cfg, self.store, self.trainable = cfg, store, trainable
        h, F = cfg['hidden_size'], cfg['ffn_dim']
        self.layers = []
        for i in range(cfg['num_layers']):
            p = f'{prefix}layers.{i}.'
            L = {}
            L['ln1w'] = store.add(p + 'self_attn_layer_norm.weight', (h,), trainable)
            L['ln1b'] = store.add(p + 'self_attn_layer_norm.bias', (h,), trainable)
            wq = store.add_fused(p + 'self_attn.qkv_fused.weight', [(p + f'self_attn.{n}_proj.weight', h, h) for n in 'qkv'], trainable)
            bq = store.add(p + 'self_attn.qkv_fused.bias', (3 * h,), trainable)
            del store.alias[bq]
            for j, n in enumerate('qkv'):
                store.alias[p + f'self_attn.{n}_proj.bias'] = (bq, j * h, (h,))
            L['qkv'] = Linear(store, wq, bq)
            L['out'] = Linear(store, store.add(p + 'self_attn.out_proj.weight', (h, h), trainable),
                              store.add(p + 'self_attn.out_proj.bias', (h,), trainable))
            L['ln2w'] = store.add(p + 'final_layer_norm.weight', (h,), trainable)
            L['ln2b'] = store.add(p + 'final_layer_norm.bias', (h,), trainable)
            L['fc1'] = Linear(store, store.add(p + 'fc1.weight', (F, h), trainable), store.add(p + 'fc1.bias', (F,), trainable))
            L['fc2'] = Linear(store, store.add(p + 'fc2.weight', (h, F), trainable), store.add(p + 'fc2.bias', (h,), trainable))
            self.layers.append(L)
        self.saved = []

    def kv_width(self):
        return 2 * self.cfg['hidden_size']

    def decode_step(self, x, cache, t, Tmax, pos, start, length):
        c, P = self.cfg, self.store.p
        h, H = c['hidden_size'], c['num_heads']
        hd = h // H
        N = x.shape[0]
        rows = torch.arange(N, device=x.device)
        for li, L in enumerate(self.layers):
            y1, _, _ = ops.layernorm_fwd(x, P[L['ln1w']], P[L['ln1b']], 1e-5, want_stats=False)
            qkv = ops.linear_small(y1, L['qkv'].w, bias=L['qkv'].b)
            cl = cache[li]
            cl.view(N, Tmax, 2 * h).index_put_((rows, t), qkv[:, h:])
            attn = ops.attn_decode(qkv[:, :h], cl, cl[:, h:], Tmax, start, length, N, H, H, hd, hd ** -0.5)
            x_mid = ops.linear_small(attn, L['out'].w, bias=L['out'].b, residual=x)
            y2, _, _ = ops.layernorm_fwd(x_mid, P[L['ln2w']], P[L['ln2b']], 1e-5, want_stats=False)
            f1 = ops.linear_small(y2, L['fc1'].w, bias=L['fc1'].b)
            a1 = ops.act_fwd(f1, ops.ACT_RELU)
            x = ops.linear_small(a1, L['fc2'].w, bias=L['fc2'].b, residual=x_mid)
        return x

    def forward(self, x, N, T, start, save, kv_sink=None):
        c, P = self.cfg, self.store.p
        h, H = c['hidden_size'], c['num_heads']
        hd = h // H
        self.saved = []
        for li, L in enumerate(self.layers):
            y1, mean1, rstd1 = ops.layernorm_fwd(x, P[L['ln1w']], P[L['ln1b']], 1e-5)
            qkv = L['qkv'].fwd(y1)
            if kv_sink is not None:
                kv_sink(li, qkv[:N * T, h:])
            attn, lse = ops.attn_fwd(qkv[:, :h], qkv[:, h:2 * h], qkv[:, 2 * h:], N, T, H, H, hd, True, hd ** -0.5, start,
                                     out=torch.zeros_like(x) if x.shape[0] != N * T else None)
            x_mid = L['out'].fwd(attn, residual=x)
            y2, mean2, rstd2 = ops.layernorm_fwd(x_mid, P[L['ln2w']], P[L['ln2b']], 1e-5)
            f1 = L['fc1'].fwd(y2)            # pre-activation kept for the ReLU backward
            a1 = ops.act_fwd(f1, ops.ACT_RELU)
            x_out = L['fc2'].fwd(a1, residual=x_mid)
            if save:
                self.saved.append((x, mean1, rstd1, y1, qkv, attn, lse, x_mid, mean2, rstd2, y2, f1, a1))
            x = x_out
        return x

    def backward(self, dres, N, T, start, on_layer_done=None):
        c, P, G = self.cfg, self.store.p, self.store.g
        h, H = c['hidden_size'], c['num_heads']
        hd = h // H
        for L, sv in zip(reversed(self.layers), reversed(self.saved)):
            x, mean1, rstd1, y1, qkv, attn, lse, x_mid, mean2, rstd2, y2, f1, a1 = sv
            d_a1 = L['fc2'].dx(dres)
            L['fc2'].dw(dres, a1)
            d_f1 = ops.act_bwd(f1, d_a1, ops.ACT_RELU)
            d_y2 = L['fc1'].dx(d_f1)
            L['fc1'].dw(d_f1, y2)
            ops.layernorm_bwd(d_y2, x_mid, P[L['ln2w']], mean2, rstd2, G.get(L['ln2w']), G.get(L['ln2b']), dx=dres, add_to_dx=True)
            d_attn = L['out'].dx(dres)
            L['out'].dw(dres, attn)
            d_qkv = torch.zeros_like(qkv) if qkv.shape[0] != N * T else torch.empty_like(qkv)
            ops.attn_bwd(qkv[:, :h], qkv[:, h:2 * h], qkv[:, 2 * h:], attn, d_attn, lse, d_qkv[:, :h], d_qkv[:, h:2 * h],
                         d_qkv[:, 2 * h:], N, T, H, H, hd, True, hd ** -0.5, start)
            d_y1 = L['qkv'].dx(d_qkv)
            L['qkv'].dw(d_qkv, y1)
            ops.layernorm_bwd(d_y1, x, P[L['ln1w']], mean1, rstd1, G.get(L['ln1w']), G.get(L['ln1b']), dx=dres, add_to_dx=True)
            if on_layer_done is not None:
                on_layer_done(L)
        self.saved = []
        return dres


class NativeOPT(NativeCausalLM):
    """hf:models/opt/modeling_opt.py OPTForCausalLM (dropout 0), lm_head tied to embed_tokens."""

    kind = 'opt'

    def __init__(self, cfg, device, trainable=True, head='lm', dtype=bf16):
        super().__init__(cfg, device, trainable, dtype)
        self.head_kind = head
        h = cfg['hidden_size']
        self.hidden_size = h
        if (h // cfg['num_heads']) not in (64, 128):
            raise NotImplementedError('OPT head_dim must be 64 or 128 for the native attention kernel')
        st = self.store
        self.embed = st.add('model.decoder.embed_tokens.weight', (cfg['vocab_size'], h), trainable, f32_grad=True)
        self.pos_emb = st.add('model.decoder.embed_positions.weight', (cfg['max_position_embeddings'] + 2, h), trainable, f32_grad=True)
        self.fln_w = st.add('model.decoder.final_layer_norm.weight', (h,), trainable)
        self.fln_b = st.add('model.decoder.final_layer_norm.bias', (h,), trainable)
        self.stack = OPTStack(cfg, st, 'model.decoder.', trainable)
        if head == 'lm':
            self.head = LMHead(st, 'ln', self.fln_w, self.fln_b, self.embed, 1e-5, trainable)
        else:  # AccustomedOPTRewardModel (models/opt.py:34-97)
            sw = st.add('score_head.weight', (1, h), trainable, f32_grad=True)
            self.head = ScoreHead(st, 'ln', self.fln_w, self.fln_b, sw, 1e-5, trainable)
        self.finalize()

    def load_state_dict(self, sd, strict=True):
        sd = dict(sd)
        sd.pop('lm_head.weight', None)  # tied
        return super().load_state_dict(sd, strict)

    def state_dict(self):
        sd = super().state_dict()
        if self.head_kind == 'lm':
            sd['lm_head.weight'] = sd['model.decoder.embed_tokens.weight']
        return sd

    def embed_tokens(self, ids, pos=None):
        P = self.store.p
        return ops.embed_fwd(ids, P[self.embed], pos=pos, P=P[self.pos_emb])

    def forward_stream(self, input_ids, attention_mask=None, pixel_values=None, save=False, image_features=None,
                       position_ids=None, kv_sink=None):
        N, T, Mp, start, _ = self._token_geometry(input_ids, attention_mask)
        P = self.store.p
        am = attention_mask if attention_mask is not None else torch.ones_like(input_ids)
        # hf:models/opt/modeling_opt.py:45-70: positions = cumsum(mask)*mask - 1 + offset(2)   (integer index work)
        am = am.to(torch.int64)
        pos = ((torch.cumsum(am, dim=1) * am - 1) + 2).to(torch.int32).reshape(-1)
        ids = input_ids.reshape(-1)
        if Mp != N * T:
            ids = torch.cat([ids, torch.zeros(Mp - N * T, dtype=ids.dtype, device=ids.device)])
            pos = torch.cat([pos, torch.zeros(Mp - N * T, dtype=torch.int32, device=ids.device)])
        x = ops.embed_fwd(ids, P[self.embed], pos=pos, P=P[self.pos_emb])
        if Mp != N * T:
            x[N * T:].zero_()
        if save:
            self._ctx = dict(ids=ids, pos=pos, N=N, T=T, start=start)
        return self.stack.forward(x, N, T, start, save, kv_sink)

    def backward_stream(self, dres, on_layer_done=None):
        cx = self._ctx
        dx = self.stack.backward(dres, cx['N'], cx['T'], cx['start'], on_layer_done)
        G = self.store.g
        if self.trainable:
            n_real = cx['N'] * cx['T']
            ops.embed_bwd(cx['ids'][:n_real], dx[:n_real], self.cfg['vocab_size'], pos=cx['pos'][:n_real],
                          dE=G[self.embed], dP=G[self.pos_emb])


def build_model(cfg: dict, device, trainable=True, head='lm', dtype=bf16, **freeze):
    """head='lm': causal LM (actor / reference); head='score': reward / critic model with a score head.
    dtype=torch.float32 selects the fp32 parity mode (weights, activations and gradients fp32)."""
    if cfg['kind'] == 'llava':
        return NativeLlava(cfg, device, trainable, head=head, dtype=dtype, **freeze)
    if cfg['kind'] == 'opt':
        return NativeOPT(cfg, device, trainable, head=head, dtype=dtype)
    if cfg['kind'] == 'llama':
        return NativeLlama(cfg, device, trainable, head=head, dtype=dtype)
    raise ValueError(f"no native model for kind {cfg['kind']!r}")
