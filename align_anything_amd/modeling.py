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
import os

import torch

from . import ops
from .params import ParamStore

bf16 = torch.bfloat16
TAIL_PRUNE = os.environ.get('AA_TAIL_PRUNE', '1') != '0'       # last decoder layer on the window rows only (LlamaStack.forward `tail`; results unchanged)


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def rope_inv_freq(hd: int, theta: float, scaling=None) -> torch.Tensor:
    """Inverse frequencies of the rotary embedding, fp32 on the host (hf:modeling_rope_utils.py ROPE_INIT_FUNCTIONS; all three types have an
    attention factor of 1, so cos / sin are not rescaled).  scaling (configs.rope_scaling_of):
      None      theta^(-2i/d)
      linear    position interpolation: every frequency divided by `factor`
      llama3    Llama-3.1: wavelengths longer than old_ctx / low_freq_factor are stretched by `factor`, those shorter than old_ctx / high_freq_factor
                are kept, the band between is blended linearly in old_ctx / wavelength."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.int64).to(torch.float32) / hd))
    if not scaling:
        return inv_freq
    if scaling['type'] == 'linear':
        return inv_freq / float(scaling['factor'])
    if scaling['type'] == 'llama3':
        factor, lo, hi = float(scaling['factor']), float(scaling['low_freq_factor']), float(scaling['high_freq_factor'])
        old_ctx = float(scaling['original_max_position_embeddings'])
        wavelen = 2.0 * math.pi / inv_freq
        stretched = torch.where(wavelen > old_ctx / lo, inv_freq / factor, inv_freq)
        blend = (old_ctx / wavelen - lo) / (hi - lo)
        mid = (1.0 - blend) * stretched / factor + blend * stretched
        in_band = ~(wavelen < old_ctx / hi) & ~(wavelen > old_ctx / lo)
        return torch.where(in_band, mid, stretched)
    raise ValueError(f'rope scaling {scaling!r} has no native implementation')


def rope_tables(max_pos: int, hd: int, theta: float, device, dtype=bf16, scaling=None):
    """hf:models/llama/modeling_llama.py:113-127: inv_freq / freqs in fp32, cos & sin cast to the activation dtype."""
    inv_freq = rope_inv_freq(hd, theta, scaling)
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
        self.tail, self.tail_used = None, False      # dead-row elimination in the last layer: see forward

    def _tables(self, T):
        if self.cos is None or self.cos.shape[0] < T:
            n = max(T, self.cfg.get('max_position_embeddings', 0) or T)
            self.cos, self.sin = rope_tables(n, self.cfg['head_dim'], self.cfg['rope_theta'], self.store.device, self.store.dtype, self.cfg.get('rope_scaling'))

    def forward(self, x, N, T, start, pos, save, kv_sink=None, tables=None, pack=None):
        """tables = (cos, sin) [rows, hd/2]: per-token rope rows (multimodal RoPE), indexed by pos[row]; default = the
        1-D position tables.
        pack (trainers.common.build_pack_plan; shared-prompt packing): x / pos are PACKED token rows (a pair's common prefix once); every row-wise
        kernel runs on them as it is, and attention runs on the reference's [N, T] layout between two row gathers (slot2row / row2slot)."""
        c, P = self.cfg, self.store.p
        H, Hkv, hd, eps = c['num_heads'], c['num_kv_heads'], c['head_dim'], c['rms_eps']
        if tables is None:
            self._tables(T)
            tables = (self.cos, self.sin)
        self._rope = tables
        self.saved = []
        qw, kw = H * hd, Hkv * hd
        # keys at or beyond kv_len[n] are masked for every query (RIGHT padding inside the sequence, as HF's padding mask does).  Unset on
        # the DPO / PPO paths: their right-padded rows are never read at a masked position.  Set by the caller (attribute `kv_len`) for the one
        # consumer that is: the vision-language reward models' end score at position -1 (models/llava.py:64-68) on a right-padded batch.
        kv_len = self._kv_len_saved = getattr(self, 'kv_len', None)
        # `tail` (set by NativeCausalLM.response_logprobs for this one call): the caller consumes the stack's output on the window rows only.  After the LAST
        # layer's keys and values nothing reads the other rows: that layer's queries, attention output, o-projection and MLP then run on the window rows alone
        # (the result comes back as [rows_pad, h] in window order).  Dead-row elimination: every consumed number is computed by the same kernels on the same
        # operands -- log-probs bit-identical; weight gradients of that layer differ by the order of fp32 partial sums only.
        tail, self.tail = self.tail, None
        if tail is not None and (kv_sink is not None or kv_len is not None or not self.layers):
            tail = None
        self._tail_saved = tail if save else None
        self.tail_used = tail is not None
        for li, L in enumerate(self.layers):
            n1, rstd1 = ops.rmsnorm_fwd(x, P[L['ln1']], eps)
            if x.dtype == bf16 and L['qkv'].b is None:
                # projection + rotary embedding in one launch (epilogue of the GEMM; falls back to the two kernels inside the library
                # when the shape does not qualify -- bit-identical either way)
                qkv = ops.gemm_qkv_rope(n1, L['qkv'].w, pos, tables[0], tables[1], H + Hkv, hd)
            else:
                qkv = L['qkv'].fwd(n1)
                ops.rope_(qkv, 0, H + Hkv, hd, pos, tables[0], tables[1])
            if kv_sink is not None:
                kv_sink(li, qkv[:N * T, qw:])  # post-RoPE keys | values of this layer -> KV cache (prefill)
            if pack is not None:
                qkv = ops.moe_gather(qkv, pack['slot2row'])          # the reference layout (pad slots: zero rows); kept for the backward instead of the packed rows
            if tail is not None and li == len(self.layers) - 1:
                attn_full, lse = ops.attn_fwd(qkv[:, :qw], qkv[:, qw:qw + kw], qkv[:, qw + kw:], N, T, H, Hkv, hd, True, hd ** -0.5, start,
                                              out=self._attn_out(qkv, N * T, H * hd), q_skip=tail['qskip'], work_frac=tail['frac'])
                attn = ops.embed_fwd(tail['gather_attn'], attn_full)  # window rows (pad rows: copies of row 0; their gradient is exactly 0)
                x_in, x = x, ops.embed_fwd(tail['gather_x'], x)
            elif pack is not None:
                attn_full, lse = ops.attn_fwd(qkv[:, :qw], qkv[:, qw:qw + kw], qkv[:, qw + kw:], N, T, H, Hkv, hd, True, hd ** -0.5, start,
                                              out=self._attn_out(qkv, N * T, H * hd), kv_len=kv_len, q_skip=pack.get('qskip'), work_frac=pack.get('attn_frac', 1.0))
                attn = ops.moe_gather(attn_full, pack['row2slot'])
            else:
                attn, lse = ops.attn_fwd(qkv[:, :qw], qkv[:, qw:qw + kw], qkv[:, qw + kw:], N, T, H, Hkv, hd, True,
                                         hd ** -0.5, start, out=self._attn_out(x, N * T, H * hd), kv_len=kv_len)
                attn_full = attn
            x_mid = L['o'].fwd(attn, residual=x)
            n2, rstd2 = ops.rmsnorm_fwd(x_mid, P[L['ln2']], eps)
            if x.dtype == bf16:
                gu, act = ops.gemm_glu_fwd(n2, L['gu'].w, c['intermediate_size'])     # [gate|up] and silu(gate)*up from one GEMM launch
            else:
                gu = L['gu'].fwd(n2)
                act = ops.swiglu_fwd(gu)
            x_out = L['down'].fwd(act, residual=x_mid)
            if save:
                if tail is not None and li == len(self.layers) - 1:
                    self.saved.append((x_in, rstd1, n1, qkv, attn, lse, x_mid, rstd2, n2, gu, act, attn_full))
                else:
                    self.saved.append((x, rstd1, n1, qkv, attn, lse, x_mid, rstd2, n2, gu, act) + ((attn_full,) if pack is not None else ()))
            x = x_out
        return x

    def kv_width(self):
        return 2 * self.cfg['num_kv_heads'] * self.cfg['head_dim']

    def prepare_decode(self, N):
        """Rollout-only copies of the layer matrices in the strip kernel's own order (ops.SwizzledWeight; +1 x the bf16 weights
        for the duration of `generate`, the weights being frozen during a rollout).  None beyond 16 sequences (tiled GEMM path)
        or with AA_DECODE_SWIZZLE=0."""
        if N > 16 or self.store.dtype != bf16 or os.environ.get('AA_DECODE_SWIZZLE', '1') == '0' or ops.DECODE_FUSED:
            return None
        # the storage persists between rollouts (same sizes every time: no allocator churn of a second copy of the model per
        # `generate`); every call refreshes it from the current weights -- one pass over the matrices, ~5 ms at 7B
        # AA_DECODE_EPI=0: plain copies + the separate SwiGLU / RoPE+cache kernels (A/B and bit-identity tests); default: the copies of gate_up
        # and (head_dim 128) qkv are row-permuted inside their strips so the strip kernel finishes those two kernels in its epilogue
        epi = os.environ.get('AA_DECODE_EPI', '1') != '0'
        # a shape the permuted copies do not fit (ffn not a multiple of 8) falls back to the plain strip copy + the separate kernel, it does not raise
        modes = {'qkv': 'rope128' if (epi and self.cfg['head_dim'] == 128) else 'plain', 'o': 'plain',
                 'gu': 'glu' if (epi and (2 * self.cfg['intermediate_size']) % 16 == 0) else 'plain', 'down': 'plain'}
        # AA_DECODE_NORM_FOLD (default on): the copies of the two projections that follow an RMSNorm are made of W diag(norm weight) and the strip
        # kernel finishes the norm itself (csrc/decode.hip PRO 4): two launches less per layer and position; 0 = the separate norm kernel
        fold = os.environ.get('AA_DECODE_NORM_FOLD', '1') != '0'
        P = self.store.p
        scale = (lambda L, k: P[L['ln1']] if k == 'qkv' else (P[L['ln2']] if k == 'gu' else None)) if fold else (lambda L, k: None)
        if getattr(self, '_dw', None) is not None and any(W[k].mode != modes[k] or W[k].folded != (scale(L, k) is not None)
                                                          for L, W in zip(self.layers[:1], self._dw[:1]) for k in modes):
            self._dw = None
        if getattr(self, '_dw', None) is None:
            self._dw = [{k: ops.SwizzledWeight(L[k].w, modes[k], kscale=scale(L, k)) for k in ('qkv', 'o', 'gu', 'down')} for L in self.layers]
        else:
            for L, W in zip(self.layers, self._dw):
                for k, sw in W.items():
                    sw.update(L[k].w)
        return self._dw

    def release_decode(self):
        """Free the rollout-only weight copies."""
        self._dw = None

    def decode_step(self, x, cache, t, Tmax, pos, start, length, dw=None):
        """One new token per sequence (x [N, h]) against the KV cache (csrc/decode.hip): every GEMM streams its
        weight once through the skinny kernel.  cache[l]: [N*Tmax, 2*kw] (keys | values), slot t is written here.
        dw: `prepare_decode` result (per-layer swizzled matrices) or None."""
        c, P = self.cfg, self.store.p
        H, Hkv, hd, eps = c['num_heads'], c['num_kv_heads'], c['head_dim'], c['rms_eps']
        qw, kw = H * hd, Hkv * hd
        N = x.shape[0]
        self._tables(Tmax)
        for li, L in enumerate(self.layers):
            # RoPE and the cache write are one pass over the new row (aa_decode_rope_cache); linear_small(norm= / swiglu=) runs the
            # RMSNorm / SwiGLU kernels itself unless ops.DECODE_FUSED folds them into the weight stream (measured slower, off)
            cl = cache[li]
            if dw is not None:       # strip-major weight copies: the element-wise kernels run on their own
                W = dw[li]
                # folded copies (W diag(norm weight)) take the residual stream itself and finish the RMSNorm in the strip kernel
                f1 = eps if W['qkv'].folded else None
                n1 = x if W['qkv'].folded else ops.rmsnorm_fwd(x, P[L['ln1']], eps)[0]
                if W['qkv'].mode == 'rope128':      # (norm +) q/k/v GEMV + RoPE + cache write in one launch
                    q = ops.gemm_skinny_rope_cache(n1, W['qkv'], L['qkv'].b, H, Hkv, pos, self.cos, self.sin, cl, Tmax, t, eps=f1)
                else:
                    qkv = ops.linear_small(n1, W['qkv'], bias=L['qkv'].b, fold_eps=f1)
                    ops.decode_rope_cache(qkv, H, Hkv, hd, pos, self.cos, self.sin, cl, Tmax, t)
                    q = qkv[:, :qw]
                attn = ops.attn_decode(q, cl, cl[:, kw:], Tmax, start, length, N, H, Hkv, hd, hd ** -0.5)
                x_mid = ops.linear_small(attn, W['o'], residual=x)
                f2 = eps if W['gu'].folded else None
                n2 = x_mid if W['gu'].folded else ops.rmsnorm_fwd(x_mid, P[L['ln2']], eps)[0]
                act = ops.gemm_skinny_glu(n2, W['gu'], eps=f2) if W['gu'].mode == 'glu' else ops.swiglu_fwd(ops.linear_small(n2, W['gu'], fold_eps=f2))
                x = ops.linear_small(act, W['down'], residual=x_mid)
                continue
            qkv = ops.linear_small(x, L['qkv'].w, bias=L['qkv'].b, norm=(P[L['ln1']], eps))
            ops.decode_rope_cache(qkv, H, Hkv, hd, pos, self.cos, self.sin, cl, Tmax, t)   # t: device int64 [N] (graph-capturable)
            attn = ops.attn_decode(qkv[:, :qw], cl, cl[:, kw:], Tmax, start, length, N, H, Hkv, hd, hd ** -0.5)
            x_mid = ops.linear_small(attn, L['o'].w, residual=x)
            gu = ops.linear_small(x_mid, L['gu'].w, norm=(P[L['ln2']], eps))
            x = ops.linear_small(gu, L['down'].w, residual=x_mid, swiglu=True)
        return x

    @staticmethod
    def _attn_out(x, real_rows, width):
        # pad rows (Mp > N*T) are never written by the attention kernel: they must read as zeros
        return None if x.shape[0] == real_rows else torch.zeros((x.shape[0], width), dtype=x.dtype, device=x.device)

    def backward(self, dres, N, T, start, pos, on_layer_done=None, pack=None):
        """dres: gradient of the residual stream after the last layer [Mp, h]; updated in place and returned
        as the gradient w.r.t. the stack input.  Weight gradients go to store.g.  pack: as in forward (pos = the full layout's positions then)."""
        c, P, G = self.cfg, self.store.p, self.store.g
        H, Hkv, hd = c['num_heads'], c['num_kv_heads'], c['head_dim']
        qw, kw = H * hd, Hkv * hd
        tr = self.trainable
        tail, self._tail_saved = getattr(self, '_tail_saved', None), None        # set: `dres` arrives as [rows_pad, h] in window order (see forward)
        for L, sv in zip(reversed(self.layers), reversed(self.saved)):
            x, rstd1, n1, qkv, attn, lse, x_mid, rstd2, n2, gu, act = sv[:11]
            attn_full = sv[11] if (pack is not None or tail is not None) else attn
            sv = None
            # ---- MLP
            if dres.dtype == bf16:
                d_gu = ops.gemm_glu_bwd(dres, L['down'].w, gu, c['intermediate_size'])   # dX GEMM of the down projection with the SwiGLU backward as epilogue
            else:
                d_gu = ops.swiglu_bwd(gu, L['down'].dx(dres))
            if tr:
                L['down'].dw(dres, act)
            d_n2 = L['gu'].dx(d_gu)
            if tr:
                L['gu'].dw(d_gu, n2)
            ops.rmsnorm_bwd(d_n2, x_mid, P[L['ln2']], rstd2, G.get(L['ln2']) if tr else None, dx=dres, add_to_dx=True)
            # ---- attention
            d_attn = L['o'].dx(dres)
            if tr:
                L['o'].dw(dres, attn)
            if tail is not None:                                      # back to the [N, T] layout / the stack's row layout: rows outside the windows have zero gradient
                d_attn = ops.moe_gather(d_attn, tail['scatter_attn'])
                dres = ops.moe_gather(dres, tail['scatter_x'])
            elif pack is not None:
                d_attn = ops.moe_gather(d_attn, pack['owner'])       # the copy of a shared prefix row in the rejected sequence is nobody's output: zero gradient
            d_qkv = torch.zeros_like(qkv) if qkv.shape[0] != N * T else torch.empty_like(qkv)
            fuse_rope = d_qkv.dtype == bf16 and ops.attn_rope_fused()      # the rotary backward rides in the dQ / dK epilogues (bit-identical)
            ops.attn_bwd(qkv[:, :qw], qkv[:, qw:qw + kw], qkv[:, qw + kw:], attn_full, d_attn, lse,
                         d_qkv[:, :qw], d_qkv[:, qw:qw + kw], d_qkv[:, qw + kw:], N, T, H, Hkv, hd, True,
                         hd ** -0.5, start, kv_len=getattr(self, '_kv_len_saved', None),
                         rope=(pos, self._rope[0], self._rope[1]) if fuse_rope else None,
                         q_skip=tail['qskip'] if tail is not None else (pack.get('qskip') if pack is not None else None),
                         work_frac=tail['frac'] if tail is not None else (pack.get('attn_frac', 1.0) if pack is not None else 1.0))
            if not fuse_rope:
                ops.rope_(d_qkv, 0, H + Hkv, hd, pos, self._rope[0], self._rope[1], inverse=True)
            if pack is not None:                                      # a shared row's gradient = the sum over its two copies (q of the second copy is exactly 0)
                d_qkv = ops.gather2_add(d_qkv, pack['row2slot'], pack['row2slot_b'])
            d_n1 = L['qkv'].dx(d_qkv)
            if tr:
                L['qkv'].dw(d_qkv, n1)
            ops.rmsnorm_bwd(d_n1, x, P[L['ln1']], rstd1, G.get(L['ln1']) if tr else None, dx=dres, add_to_dx=True)
            tail = None                                               # the last layer only
            if on_layer_done is not None:
                on_layer_done(L)
        self.saved = []
        return dres


# ====================================================================== CLIP vision tower (frozen: forward only)
class ClipVisionTower:
    """hf:models/clip/modeling_clip.py:138-218, :338-351, :605-651.  Runs layers 0..L+feature_layer only (the
    reference computes the last block(s) and post_layernorm and discards them, modeling_llava.py:154-166)."""

    def __init__(self, vcfg: dict, store: ParamStore, prefix: str, feature_layer: int = -2, trainable: bool = False):
        self.cfg, self.store, self.prefix, self.trainable = vcfg, store, prefix, trainable
        h, F = vcfg['hidden_size'], vcfg['intermediate_size']
        self.K = vcfg['num_channels'] * vcfg['patch_size'] ** 2
        self.Kp = _pad64(self.K)
        self.G2 = (vcfg['image_size'] // vcfg['patch_size']) ** 2
        if h // vcfg['num_heads'] not in (64, 128):
            raise NotImplementedError('CLIP head_dim must be 64 or 128 for the native attention kernel')
        nl = vcfg['num_layers']
        self.run_layers = nl + 1 + feature_layer if feature_layer < 0 else feature_layer  # hidden_states index
        tr = trainable
        e = prefix + 'embeddings.'
        self.patch_w = store.add(e + 'patch_embedding.weight', (h, self.Kp), tr)
        self.cls = store.add(e + 'class_embedding', (h,), tr)
        self.pos = store.add(e + 'position_embedding.weight', (self.G2 + 1, h), tr, f32_grad=True)   # accumulated by a column sum
        self.pre_w = store.add(prefix + 'pre_layrnorm.weight', (h,), tr)
        self.pre_b = store.add(prefix + 'pre_layrnorm.bias', (h,), tr)
        self.layers = []
        for i in range(vcfg['num_layers']):
            p = f'{prefix}encoder.layers.{i}.'
            L = {}
            tr = trainable and i < self.run_layers          # blocks past the feature layer get no gradient in HF either
            L['ln1w'] = store.add(p + 'layer_norm1.weight', (h,), tr)
            L['ln1b'] = store.add(p + 'layer_norm1.bias', (h,), tr)
            wq = store.add_fused(p + 'self_attn.qkv_fused.weight', [(p + f'self_attn.{n}_proj.weight', h, h) for n in 'qkv'], tr)
            # biases are 1-D: fuse by registering one [3h] block with three aliases
            bq = store.add(p + 'self_attn.qkv_fused.bias', (3 * h,), tr)
            del store.alias[bq]
            for j, n in enumerate('qkv'):
                store.alias[p + f'self_attn.{n}_proj.bias'] = (bq, j * h, (h,))
            L['qkv'] = Linear(store, wq, bq)
            L['out'] = Linear(store, store.add(p + 'self_attn.out_proj.weight', (h, h), tr),
                              store.add(p + 'self_attn.out_proj.bias', (h,), tr))
            L['ln2w'] = store.add(p + 'layer_norm2.weight', (h,), tr)
            L['ln2b'] = store.add(p + 'layer_norm2.bias', (h,), tr)
            L['fc1'] = Linear(store, store.add(p + 'mlp.fc1.weight', (F, h), tr), store.add(p + 'mlp.fc1.bias', (F,), tr))
            L['fc2'] = Linear(store, store.add(p + 'mlp.fc2.weight', (h, F), tr), store.add(p + 'mlp.fc2.bias', (h,), tr))
            self.layers.append(L)
        self.post_w = store.add(prefix + 'post_layernorm.weight', (h,), False)
        self.post_b = store.add(prefix + 'post_layernorm.bias', (h,), False)
        self._drop_cls_idx = {}
        self._ctx = None

    def forward(self, pixel_values, save=False):
        """pixel_values [n, 3, S, S] (fp32 or bf16) -> patch features [n * G2, h] (CLS dropped)."""
        c, P = self.cfg, self.store.p
        n = pixel_values.shape[0]
        h, H = c['hidden_size'], c['num_heads']
        hd, eps, T = h // H, c['ln_eps'], self.G2 + 1
        keep = save and self.trainable
        col = ops.patch_im2col(pixel_values.contiguous(), c['patch_size'], self.Kp, self.store.dtype)
        if keep and col.shape[0] % 64:      # training: every row count is the contraction dim of a dW GEMM
            col = torch.cat([col, torch.zeros((_pad64(col.shape[0]) - col.shape[0], self.Kp), dtype=col.dtype, device=col.device)])
        pe = ops.gemm(col, P[self.patch_w])
        x0 = ops.clip_embed(pe, P[self.cls], P[self.pos], n, self.G2)
        M = n * T
        if keep and M % 64:
            x0 = torch.cat([x0, torch.zeros((_pad64(M) - M, h), dtype=x0.dtype, device=x0.device)])
        x, mp, rp = ops.layernorm_fwd(x0, P[self.pre_w], P[self.pre_b], eps, want_stats=keep)
        saved = []
        for L in self.layers[:self.run_layers]:
            y1, m1, r1 = ops.layernorm_fwd(x, P[L['ln1w']], P[L['ln1b']], eps, want_stats=keep)
            qkv = L['qkv'].fwd(y1)
            a, lse = ops.attn_fwd(qkv[:, :h], qkv[:, h:2 * h], qkv[:, 2 * h:], n, T, H, H, hd, False, hd ** -0.5,
                                  out=torch.zeros_like(x) if x.shape[0] != M else None)
            x_mid = L['out'].fwd(a, residual=x)
            y2, m2, r2 = ops.layernorm_fwd(x_mid, P[L['ln2w']], P[L['ln2b']], eps, want_stats=keep)
            if keep:
                f1 = L['fc1'].fwd(y2)
                g1 = ops.act_fwd(f1, ops.ACT_QUICK_GELU)
                saved.append((x, m1, r1, y1, qkv, a, lse, x_mid, m2, r2, y2, f1, g1))
            else:
                g1 = L['fc1'].fwd(y2, act=ops.ACT_QUICK_GELU)
            x = L['fc2'].fwd(g1, residual=x_mid)
        idx = self._drop_cls_idx.get(n)
        if idx is None:
            idx = (torch.arange(n * T).view(n, T)[:, 1:]).reshape(-1).to(self.store.device)
            self._drop_cls_idx[n] = idx
        if keep:
            self._ctx = dict(n=n, T=T, M=M, rows=x.shape[0], col=col, x0=x0, mp=mp, rp=rp, saved=saved, idx=idx)
        return ops.embed_fwd(idx, x)  # row gather: drop the CLS token of every image

    def backward(self, dfeat):
        """dfeat [>= n * G2, h]: gradient of the returned patch features.  Accumulates the gradients of the blocks up to the feature
        layer, pre_layrnorm and the embeddings (patch conv, class token, positions) into store.g."""
        cx, c, P, G = self._ctx, self.cfg, self.store.p, self.store.g
        n, T, M, rows = cx['n'], cx['T'], cx['M'], cx['rows']
        h, H = c['hidden_size'], c['num_heads']
        hd = h // H
        dres = torch.zeros((rows, h), dtype=dfeat.dtype, device=dfeat.device)
        dres.index_copy_(0, cx['idx'], dfeat[:cx['idx'].numel()])          # CLS rows and pad rows get no gradient from the features
        for L, sv in zip(reversed(self.layers[:self.run_layers]), reversed(cx['saved'])):
            x, m1, r1, y1, qkv, a, lse, x_mid, m2, r2, y2, f1, g1 = sv
            d_g1 = L['fc2'].dx(dres)
            L['fc2'].dw(dres, g1)
            d_f1 = ops.act_bwd(f1, d_g1, ops.ACT_QUICK_GELU)
            d_y2 = L['fc1'].dx(d_f1)
            L['fc1'].dw(d_f1, y2)
            ops.layernorm_bwd(d_y2, x_mid, P[L['ln2w']], m2, r2, G.get(L['ln2w']), G.get(L['ln2b']), dx=dres, add_to_dx=True)
            d_a = L['out'].dx(dres)
            L['out'].dw(dres, a)
            d_qkv = torch.zeros_like(qkv) if rows != M else torch.empty_like(qkv)
            ops.attn_bwd(qkv[:, :h], qkv[:, h:2 * h], qkv[:, 2 * h:], a, d_a, lse, d_qkv[:, :h], d_qkv[:, h:2 * h], d_qkv[:, 2 * h:],
                         n, T, H, H, hd, False, hd ** -0.5)
            d_y1 = L['qkv'].dx(d_qkv)
            L['qkv'].dw(d_qkv, y1)
            ops.layernorm_bwd(d_y1, x, P[L['ln1w']], m1, r1, G.get(L['ln1w']), G.get(L['ln1b']), dx=dres, add_to_dx=True)
        d_x0 = ops.layernorm_bwd(dres, cx['x0'], P[self.pre_w], cx['mp'], cx['rp'], G.get(self.pre_w), G.get(self.pre_b))
        # embeddings (modeling_clip.py:190-218): x0[n, 0] = cls + pos[0], x0[n, 1+p] = patch[n, p] + pos[1+p]
        dpos = G[self.pos]                                                   # fp32 [T, h], accumulated
        ops.colsum_(d_x0[:M].view(n, T * h), dpos.view(-1))
        dcls_src = torch.zeros(h, dtype=torch.float32, device=dres.device)
        cls_rows = torch.arange(n, device=dres.device) * T
        ops.colsum_(ops.embed_fwd(cls_rows, d_x0), dcls_src)
        G[self.cls].add_(dcls_src)
        d_pe = torch.zeros((cx['col'].shape[0], h), dtype=d_x0.dtype, device=dres.device)
        d_pe[:cx['idx'].numel()] = ops.embed_fwd(cx['idx'], d_x0)
        Linear(self.store, self.patch_w).dw(d_pe, cx['col'])
        self._ctx = None


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
        if ops.lmhead_fused():      # lm_head inside the log-prob pass: no [rows, V] buffer, backward recomputes the chunks
            logp, lse = ops.lmhead_logprob_fwd(n, self._w(), labels, round_bf16)
            logits = None
        else:
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

    def prepare_decode(self, N):
        """lm_head in the strip kernel's order for the rollout (see LlamaStack.prepare_decode)."""
        if N > 16 or self.store.dtype != bf16 or os.environ.get('AA_DECODE_SWIZZLE', '1') == '0' or ops.DECODE_FUSED:
            return None
        fold = self.kind == 'rms' and os.environ.get('AA_DECODE_NORM_FOLD', '1') != '0'      # the final RMSNorm folded into the lm_head copy
        if getattr(self, '_dw', None) is not None and self._dw.folded != fold:
            self._dw = None
        if getattr(self, '_dw', None) is None:
            self._dw = ops.SwizzledWeight(self._w(), kscale=self.store.p[self.norm_w] if fold else None)
        else:
            self._dw.update(self._w())
        return self._dw

    def logits_rows(self, x_rows, w=None, out=None):
        """Logits of a handful of rows (decode): norm + skinny lm_head (w: `prepare_decode` result or None; out: written in place when given)."""
        P = self.store.p
        if w is not None and getattr(w, 'folded', False):
            return ops.linear_small(x_rows, w, fold_eps=self.eps, out=out)
        if out is not None:
            return out.copy_(self.logits_rows(x_rows, w))
        if self.kind == 'rms':
            if x_rows.dtype == bf16 and w is None:
                return ops.linear_small(x_rows, self._w(), norm=(P[self.norm_w], self.eps))     # (norm folded in with ops.DECODE_FUSED)
            n, _ = ops.rmsnorm_fwd(x_rows, P[self.norm_w], self.eps)
        else:
            n, _, _ = ops.layernorm_fwd(x_rows, P[self.norm_w], P[self.norm_b], self.eps, want_stats=False)
        return ops.linear_small(n, self._w() if w is None else w)

    def hidden_all(self, x_last):
        P = self.store.p
        if self.kind == 'rms':
            return ops.rmsnorm_fwd(x_last, P[self.norm_w], self.eps)[0]
        return ops.layernorm_fwd(x_last, P[self.norm_w], P[self.norm_b], self.eps, want_stats=False)[0]

    def backward(self, dlogp, inv_map, zero_row, compact=False):
        """dlogp f32[rows_pad] -> gradient of the residual stream [Mp, h] (rows outside the windows are 0); compact: [rows_pad, h] in window order instead."""
        sel, mean, rstd, n, logits, lse, labels = self.saved
        P, G = self.store.p, self.store.g
        gw = None
        if self.trainable:
            gw = G.get(self.lm_w)
            if gw is None:
                gw = self.store.grad_view(self.lm_w)
        acc = gw is not None and ((gw.dtype == torch.float32) or self.store.accumulate)
        if logits is None:
            d_n = ops.lmhead_logprob_bwd(n, self._w(), labels, lse, dlogp, dw=gw, accumulate=acc)
        else:
            ops.logprob_gather_bwd(logits, labels, lse, dlogp, out=logits)  # in place: logits -> dlogits
            d_n = ops.gemm(logits, self._w(), b_n=True)
            if gw is not None:
                ops.gemm(logits, n, out=gw, a_t=True, b_n=True, accumulate=acc)
        tr = self.trainable
        if self.kind == 'rms':
            d_sel = ops.rmsnorm_bwd(d_n, sel, P[self.norm_w], rstd, G.get(self.norm_w) if tr else None)
        else:
            d_sel = ops.layernorm_bwd(d_n, sel, P[self.norm_w], mean, rstd, G.get(self.norm_w) if tr else None,
                                      G.get(self.norm_b) if tr else None)
        self.saved = None
        if compact:
            return d_sel
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

    def backward(self, dscores, inv_map, zero_row, compact=False):
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
        if compact:
            return d_sel
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

    def decode_start_positions(self, valid):
        """RoPE position of the first generated token of every row (generation.py); HF: number of attended tokens."""
        return valid

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
                          image_features=None, round_bf16=False, pack=None, **mm):
        """window: dict(row_idx int64[rows_pad], labels int64[rows_pad], inv_map int32[Mp]) built by
        trainers.common.build_window.  Returns flat fp32 log-probs [rows_pad] (pad rows meaningless).
        pack: trainers.common.build_pack_plan (shared-prompt packing; its own `window` indexes the packed rows)."""
        if pack is not None:
            window = pack['window']
            mm = dict(mm, pack=pack)
        for k in ('row_idx', 'labels', 'inv_map'):      # a host-side plan handed to the kernels would be a wild device pointer
            if window[k].device.type != self.device.type:
                raise RuntimeError(f'response_logprobs: window[{k!r}] lives on {window[k].device}, the model on {self.device}')
        x, compact = self._forward_to_window(window, pack, mm, input_ids, attention_mask, pixel_values, save, image_features)
        logp = self.head.forward(x, window['row_idx_id'] if compact else window['row_idx'], window['labels'], save, round_bf16)
        if save:
            self._ctx['window'] = window
            self._ctx['tail'] = compact
        return logp

    def _forward_to_window(self, window, pack, mm, input_ids, attention_mask, pixel_values, save, image_features):
        """forward_stream for a caller that reads the window rows only -> (x, compact).  Dead-row elimination in the last decoder layer (LlamaStack.forward,
        `tail`): with compact the result is [rows_pad, h] in window order.  AA_TAIL_PRUNE=0: off (A/B)."""
        stack = getattr(self, 'stack', None)
        want_tail = (TAIL_PRUNE and 'tail_qskip' in window and hasattr(stack, 'tail_used') and (pack is None or 'tail' in pack)
                     and 'position_ids' not in mm and 'kv_sink' not in mm and mm.get('kv_len') is None)
        if want_tail:       # gather_*: window row -> row of the attention output ([N, T] layout) / of the stack's rows; scatter_*: the inverse maps (-1: not a window row)
            stack.tail = pack['tail'] if pack is not None else {
                'gather_attn': window['tail_gather'], 'gather_x': window['tail_gather'], 'scatter_attn': window['inv_map'], 'scatter_x': window['inv_map'],
                'qskip': window['tail_qskip'], 'frac': window['tail_frac']}
        try:
            x = self.forward_stream(input_ids, attention_mask, pixel_values, save, image_features, **mm)
        finally:
            if want_tail:
                stack.tail = None
        return x, bool(want_tail and stack.tail_used)

    def response_scores(self, input_ids, attention_mask, window, pixel_values=None, save=False, image_features=None,
                        all_scores=False, **mm):
        """Score-head models: fp32 scores on the window rows (critic values / reward scores).  all_scores=True additionally
        returns ScoreModelOutput.scores [N, T] of the same forward (no gradient flows through it): what the reference's
        RMTrainer.loss reports as higher_rewards / lower_rewards (trainers/text_to_text/rm.py:110-130)."""
        if all_scores:
            x, compact = self.forward_stream(input_ids, attention_mask, pixel_values, save, image_features, **mm), False
        else:       # critic values / reward scores on the window rows: nothing else of the last layer is read
            x, compact = self._forward_to_window(window, None, mm, input_ids, attention_mask, pixel_values, save, image_features)
        sc = self.head.forward(x, window['row_idx_id'] if compact else window['row_idx'], None, save)
        if save:
            self._ctx['window'] = window
            self._ctx['tail'] = compact
        if all_scores:
            N, T = input_ids.shape
            return sc, self.head.scores_all(x)[:N * T].view(N, T)
        return sc

    def scores(self, input_ids, attention_mask=None, pixel_values=None, **mm):
        """ScoreModelOutput.scores [N, T] (fp32) for every position."""
        N, T = input_ids.shape
        x = self.forward_stream(input_ids, attention_mask, pixel_values, save=False, **mm)
        return self.head.scores_all(x)[:N * T].view(N, T)

    def backward_from_dlogp(self, dlogp, on_layer_done=None):
        dres = self.head.backward(dlogp, self._ctx['window']['inv_map'], self._zero_row, compact=bool(self._ctx.get('tail', False)))
        self.backward_stream(dres, on_layer_done)
        self._ctx = None

    def logits(self, input_ids, attention_mask=None, pixel_values=None, **mm):
        """All-position logits [N, T, V] (what HF returns) -- parity tests and PPO/generation callers."""
        N, T = input_ids.shape
        x = self.forward_stream(input_ids, attention_mask, pixel_values, save=False, **mm)
        return self.head.logits_all(x)[:N * T].view(N, T, -1)

    def final_hidden(self, input_ids, attention_mask=None, pixel_values=None, **mm):
        N, T = input_ids.shape
        x = self.forward_stream(input_ids, attention_mask, pixel_values, save=False, **mm)
        return self.head.hidden_all(x)[:N * T].view(N, T, -1)


# ====================================================================== LLaVA
class NativeLlava(NativeCausalLM):
    """hf:models/llava/modeling_llava.py:301-371 LlavaForConditionalGeneration, natively."""

    kind = 'llava'

    def __init__(self, cfg, device, trainable=True, freeze_mm_proj=False, freeze_language_model=False,
                 freeze_vision_tower=True, head='lm', dtype=bf16):
        super().__init__(cfg, device, trainable, dtype)
        self.head_kind = head
        t = cfg['text']
        self.hidden_size = t['hidden_size']
        self.train_lm = trainable and not freeze_language_model
        self.train_proj = trainable and not freeze_mm_proj
        self.train_tower = trainable and not freeze_vision_tower     # reference default: frozen (configs/train/text_image_to_text/dpo.yaml:60)
        st = self.store
        self.vision = ClipVisionTower(cfg['vision'], st, 'model.vision_tower.', cfg.get('vision_feature_layer', -2), self.train_tower)
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
                       position_ids=None, kv_sink=None, kv_len=None, pack=None):
        """kv_len (int32 [N], optional): keys at or beyond it are masked in the decoder (right padding; see LlamaStack.forward).
        pack: shared-prompt packing plan (trainers.common.build_pack_plan): token ids / positions / image features are per PACKED row -- one set of image
        features per pair."""
        N, T, Mp, start, pos = self._token_geometry(input_ids, attention_mask, position_ids)
        self.stack.kv_len = kv_len
        P = self.store.p
        ids = input_ids.reshape(-1)
        if pack is not None:
            if position_ids is not None or kv_sink is not None:
                raise RuntimeError('shared-prompt packing is a training-forward layout (no explicit position ids, no KV-cache prefill)')
            ids, pos = pack['ids'], pack['pos']
        elif Mp != N * T:
            ids = torch.cat([ids, torch.zeros(Mp - N * T, dtype=ids.dtype, device=ids.device)])
        slot = feat = f1 = a1 = vfeat = None
        if pixel_values is not None or image_features is not None:
            vfeat = image_features if image_features is not None else self.vision.forward(pixel_values, save=save)
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
            self._ctx = dict(ids=ids, slot=slot, f1=f1, a1=a1, vfeat=vfeat, N=N, T=T, start=start, pos=pos if pack is None else pack['pos_full'], pack=pack)
        return self.stack.forward(x, N, T, start, pos, save, kv_sink, pack=pack)

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
        dx = self.stack.backward(dres, cx['N'], cx['T'], cx['start'], cx['pos'], on_layer_done, pack=cx.get('pack'))
        G = self.store.g
        tower = self.train_tower and self.vision._ctx is not None
        want_feat = cx['slot'] is not None and (self.train_proj or tower)
        dfeat = torch.zeros((cx['vfeat'].shape[0], self.hidden_size), dtype=self.dtype, device=self.device) if want_feat else None
        if self.train_lm or want_feat:
            ops.embed_bwd(cx['ids'], dx, self.cfg['text']['vocab_size'], slot=cx['slot'],
                          dE=G.get(self.embed) if self.train_lm else None, dfeat=dfeat)
        if want_feat:
            if self.train_proj:
                self.proj2.dw(dfeat, cx['a1'])
            d_a1 = self.proj2.dx(dfeat)
            d_f1 = ops.act_bwd(cx['f1'], d_a1, ops.ACT_GELU)
            if self.train_proj:
                self.proj1.dw(d_f1, cx['vfeat'])
            if tower:
                self.vision.backward(self.proj1.dx(d_f1))


# ====================================================================== Qwen2-VL
def qwen2vl_rope_index(input_ids, attention_mask, grid_thw, image_token_id: int, merge: int):
    """hf:models/qwen2_vl/modeling_qwen2_vl.py:862-1018 get_rope_index (+ get_vision_position_ids) on the host: 3-D position
    ids int32 [3, N, T] (pad positions 0) and the per-row rope deltas (max position + 1 - attended length).  Integer
    work -- pinned bit-exactly to HF through tests/golden/qwen2vl_tiny_dpo.npz.  Token types come from
    input_ids == image_token_id (what the processor's mm_token_type_ids encodes)."""
    import numpy as np
    ids = input_ids.cpu().numpy() if isinstance(input_ids, torch.Tensor) else np.asarray(input_ids)
    N, T = ids.shape
    am = np.ones((N, T), dtype=bool) if attention_mask is None else (
        attention_mask.cpu().numpy() if isinstance(attention_mask, torch.Tensor) else np.asarray(attention_mask)).astype(bool)
    pos = np.zeros((3, N, T), dtype=np.int32)
    deltas = np.zeros(N, dtype=np.int32)
    gi = 0
    for b in range(N):
        keep = np.nonzero(am[b])[0]
        types = ids[b, keep] == image_token_id
        out = np.zeros((3, len(keep)), dtype=np.int64)
        cur, i, n = 0, 0, len(keep)
        while i < n:
            j = i
            while j < n and types[j] == types[i]:
                j += 1
            if not types[i]:
                out[:, i:j] = np.arange(j - i)[None] + cur
                cur += j - i
            else:
                t, h, w = (int(v) for v in grid_thw[gi]); gi += 1
                gh, gw = h // merge, w // merge
                if t * gh * gw != j - i:
                    raise ValueError(f'row {b}: {j - i} image tokens for a {t}x{h}x{w} grid (expected {t * gh * gw})')
                tt, hh, ww = np.meshgrid(np.arange(t), np.arange(gh) + cur, np.arange(gw) + cur, indexing='ij')
                out[0, i:j] = tt.reshape(-1) + cur; out[1, i:j] = hh.reshape(-1); out[2, i:j] = ww.reshape(-1)
                cur += max(h, w) // merge
            i = j
        pos[:, b, keep] = out
        deltas[b] = (out.max() + 1 - n) if n else 0
    return pos, deltas


class Qwen2VLVisionTower:
    """hf:models/qwen2_vl/modeling_qwen2_vl.py:649-730 Qwen2VisionTransformerPretrainedModel, forward and (train_blocks) backward:
    patch embed (Conv3d with kernel == stride = one GEMM over the processor's flattened patches), `depth` pre-LN blocks
    with 2-D rotary (fp32, :225-236) and full attention inside every (image, frame) segment, then the 2x2 PatchMerger.
    head_dim 80 (1280 / 16) has no attention kernel: q/k/v heads are produced zero-padded to 128 columns by padding the
    ROWS of the qkv weight (and the columns of the output projection) -- the scores and the output are unchanged, the softmax
    scale stays head_dim**-0.5; when the blocks train, the padded copies are rebuilt every forward and the gradients of the
    padded matrices are copied back into the HF-shaped gradient buffers."""

    def __init__(self, vcfg: dict, store: ParamStore, prefix: str, train_merger: bool, train_blocks: bool = False):
        self.cfg, self.store, self.prefix, self.train_merger, self.train_blocks = vcfg, store, prefix, train_merger, train_blocks
        E, H = vcfg['embed_dim'], vcfg['num_heads']
        self.hd = E // H
        if self.hd > 128 or self.hd % 16:
            raise NotImplementedError(f'vision head_dim {self.hd}: must be a multiple of 16 and <= 128')
        self.hdp = self.hd if self.hd in (64, 128) else 128
        self.K = vcfg['in_channels'] * vcfg['temporal_patch_size'] * vcfg['patch_size'] ** 2
        self.Kp = _pad64(self.K)
        F = int(E * vcfg['mlp_ratio'])
        m2 = vcfg['spatial_merge_size'] ** 2
        tb = train_blocks
        self.patch_w = store.add(prefix + 'patch_embed.proj.weight', (E, self.Kp), tb)
        self.blocks = []
        for i in range(vcfg['depth']):
            p = f'{prefix}blocks.{i}.'
            B = {'n1w': store.add(p + 'norm1.weight', (E,), tb), 'n1b': store.add(p + 'norm1.bias', (E,), tb),
                 'n2w': store.add(p + 'norm2.weight', (E,), tb), 'n2b': store.add(p + 'norm2.bias', (E,), tb),
                 'qkv_w': store.add(p + 'attn.qkv.weight', (3 * E, E), tb), 'qkv_b': store.add(p + 'attn.qkv.bias', (3 * E,), tb),
                 'proj_w': store.add(p + 'attn.proj.weight', (E, E), tb), 'proj_b': store.add(p + 'attn.proj.bias', (E,), tb),
                 'fc1': Linear(store, store.add(p + 'mlp.fc1.weight', (F, E), tb), store.add(p + 'mlp.fc1.bias', (F,), tb)),
                 'fc2': Linear(store, store.add(p + 'mlp.fc2.weight', (E, F), tb), store.add(p + 'mlp.fc2.bias', (E,), tb))}
            self.blocks.append(B)
        tm = train_merger
        self.lnq_w = store.add(prefix + 'merger.ln_q.weight', (E,), tm)
        self.lnq_b = store.add(prefix + 'merger.ln_q.bias', (E,), tm)
        self.m0 = Linear(store, store.add(prefix + 'merger.mlp.0.weight', (E * m2, E * m2), tm), store.add(prefix + 'merger.mlp.0.bias', (E * m2,), tm))
        self.m2 = Linear(store, store.add(prefix + 'merger.mlp.2.weight', (vcfg['hidden_size'], E * m2), tm),
                         store.add(prefix + 'merger.mlp.2.bias', (vcfg['hidden_size'],), tm))
        self._padded = None
        self._ctx = None

    def invalidate(self):
        self._padded = None

    def _attn_weights(self):
        """Per block (qkv_w, qkv_b, proj_w): the HF tensors when head_dim has a kernel, else head-padded copies."""
        if self._padded is None:
            P, E, H, hd, hdp = self.store.p, self.cfg['embed_dim'], self.cfg['num_heads'], self.hd, self.hdp
            out = []
            for B in self.blocks:
                if hd == hdp:
                    out.append((P[B['qkv_w']], P[B['qkv_b']], P[B['proj_w']]))
                    continue
                qw = torch.zeros((3, H, hdp, E), dtype=self.store.dtype, device=self.store.device)
                qw[:, :, :hd] = P[B['qkv_w']].view(3, H, hd, E)
                qb = torch.zeros((3, H, hdp), dtype=self.store.dtype, device=self.store.device)
                qb[:, :, :hd] = P[B['qkv_b']].view(3, H, hd)
                pw = torch.zeros((E, H, hdp), dtype=self.store.dtype, device=self.store.device)
                pw[:, :, :hd] = P[B['proj_w']].view(E, H, hd)
                out.append((qw.view(3 * H * hdp, E), qb.view(-1), pw.view(E, H * hdp)))
            self._padded = out
        return self._padded

    def _rotary(self, grid_thw):
        """hf:vision_utils.py get_vision_position_ids + VisionRotaryEmbedding: fp32 cos/sin rows [n_patches, head_dim/2]
        (block-major h / w ids of the 2x2 merge groups), and the attention segments [(row0, n_seq, seq_len)]."""
        import numpy as np
        m, hd = self.cfg['spatial_merge_size'], self.hd
        pos, segs, o = [], [], 0
        for t, h, w in grid_thw:
            hp, wp = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')
            blk = (h // m, m, w // m, m)
            hp = hp.reshape(blk).transpose(0, 2, 1, 3).reshape(-1)
            wp = wp.reshape(blk).transpose(0, 2, 1, 3).reshape(-1)
            pos.append(np.tile(np.stack([hp, wp], -1), (t, 1)))
            segs.append((o, t, h * w)); o += t * h * w
        pos = torch.from_numpy(np.concatenate(pos, 0)).to(torch.float32)
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float32) / (hd // 2)))
        freqs = (pos.unsqueeze(-1) * inv_freq).flatten(1)             # [n, hd/2]  (h part | w part)
        return freqs.cos().contiguous(), freqs.sin().contiguous(), segs, o

    def forward(self, pixel_values, grid_thw, save=False):
        """pixel_values [n_patches, C*tps*ps*ps] (fp32 / bf16), grid_thw [[t, h, w], ...] host ints -> merged features
        [n_patches / merge^2 (padded to 64 rows, zero), hidden_size]."""
        c, P, dev, dt = self.cfg, self.store.p, self.store.device, self.store.dtype
        grid = [[int(v) for v in g] for g in (grid_thw.tolist() if isinstance(grid_thw, torch.Tensor) else grid_thw)]
        E, H, hd, hdp = c['embed_dim'], c['num_heads'], self.hd, self.hdp
        cos, sin, segs, n = self._rotary(grid)
        if pixel_values.shape[0] != n or pixel_values.shape[1] != self.K:
            raise ValueError(f'pixel_values {tuple(pixel_values.shape)} does not match image_grid_thw ({n} patches of {self.K})')
        cos, sin = cos.to(dev), sin.to(dev)
        keep = save and self.train_blocks
        npad = _pad64(n) if keep else n                     # training: rows are the contraction dim of the dW GEMMs
        pix = torch.zeros((npad, self.Kp), dtype=dt, device=dev)
        pix[:n, :self.K] = pixel_values.to(dt)
        x = ops.gemm(pix, P[self.patch_w])
        rows = torch.arange(npad, dtype=torch.int32, device=dev)
        if npad != n:
            cos = torch.cat([cos, torch.ones((npad - n, cos.shape[1]), device=dev)]); sin = torch.cat([sin, torch.zeros((npad - n, sin.shape[1]), device=dev)])
        if self.train_blocks:
            self._padded = None                             # the weights moved since the last step
        W = self._attn_weights()
        saved = []
        for B, (qkv_w, qkv_b, proj_w) in zip(self.blocks, W):
            y1, m1, r1 = ops.layernorm_fwd(x, P[B['n1w']], P[B['n1b']], 1e-6, want_stats=keep)
            qkv = ops.gemm(y1, qkv_w, bias=qkv_b)
            ops.rope_(qkv, 0, 2 * H, hd, rows, cos, sin, head_stride=hdp, precise=True)
            a = torch.zeros((npad, H * hdp), dtype=dt, device=dev) if npad != n else torch.empty((n, H * hdp), dtype=dt, device=dev)
            lses = []
            for o, nseq, L in segs:
                v = slice(o, o + nseq * L)
                _, lse = ops.attn_fwd(qkv[v, :H * hdp], qkv[v, H * hdp:2 * H * hdp], qkv[v, 2 * H * hdp:], nseq, L, H, H, hdp, False, hd ** -0.5,
                                      out=a[v])
                lses.append(lse)
            x_mid = ops.gemm(a, proj_w, bias=P[B['proj_b']], residual=x)
            y2, m2_, r2 = ops.layernorm_fwd(x_mid, P[B['n2w']], P[B['n2b']], 1e-6, want_stats=keep)
            if keep:
                f1 = B['fc1'].fwd(y2)
                g1 = ops.act_fwd(f1, ops.ACT_QUICK_GELU)
                saved.append((x, m1, r1, y1, qkv, a, lses, x_mid, m2_, r2, y2, f1, g1, qkv_w, proj_w))
            else:
                g1 = B['fc1'].fwd(y2, act=ops.ACT_QUICK_GELU)
            x = B['fc2'].fwd(g1, residual=x_mid)
        x = x[:n] if npad != n else x
        m2 = c['spatial_merge_size'] ** 2
        y, mean, rstd = ops.layernorm_fwd(x, P[self.lnq_w], P[self.lnq_b], 1e-6)
        nf = n // m2
        y4 = torch.zeros((_pad64(nf), E * m2), dtype=dt, device=dev)      # pad rows zero: K of the merger dW GEMMs
        y4[:nf] = y.view(nf, E * m2)
        f1 = self.m0.fwd(y4)
        a1 = ops.act_fwd(f1, ops.ACT_GELU)
        feat = self.m2.fwd(a1)
        if save:
            self._ctx = dict(x=x, mean=mean, rstd=rstd, y4=y4, f1=f1, a1=a1, nf=nf, n=n, npad=npad, saved=saved, pix=pix, segs=segs,
                             cos=cos, sin=sin, rows=rows)
        return feat, nf

    def backward_merger(self, dfeat):
        """Gradients of the PatchMerger (ln_q, mlp.0, mlp.2), then of the blocks and the patch embedding when they train."""
        cx, G, P = self._ctx, self.store.g, self.store.p
        self.m2.dw(dfeat, cx['a1'])
        d_a1 = self.m2.dx(dfeat)
        d_f1 = ops.act_bwd(cx['f1'], d_a1, ops.ACT_GELU)
        self.m0.dw(d_f1, cx['y4'])
        d_y4 = self.m0.dx(d_f1)
        E = self.cfg['embed_dim']
        d_y = d_y4[:cx['nf']].reshape(-1, E).contiguous()
        d_x = ops.layernorm_bwd(d_y, cx['x'].contiguous(), P[self.lnq_w], cx['mean'], cx['rstd'], G.get(self.lnq_w), G.get(self.lnq_b))
        if self.train_blocks:
            self._backward_blocks(d_x)
        self._ctx = None

    def _unpad_grad(self, name, tmp, view_pad, view_real, sel):
        """Copy the gradient of a head-padded matrix into the HF-shaped gradient buffer (add when accumulating)."""
        g = self.store.g[name]
        src = tmp.view(*view_pad)[sel].to(g.dtype)
        if g.dtype == torch.float32 or self.store.accumulate:
            g.view(*view_real).add_(src)
        else:
            g.view(*view_real).copy_(src)

    def _backward_blocks(self, d_x):
        cx, c, P, G, st = self._ctx, self.cfg, self.store.p, self.store.g, self.store
        E, H, hd, hdp, n, npad = c['embed_dim'], c['num_heads'], self.hd, self.hdp, cx['n'], cx['npad']
        dres = d_x
        if npad != n:
            dres = torch.zeros((npad, E), dtype=d_x.dtype, device=d_x.device); dres[:n] = d_x
        acc = lambda g: (g.dtype == torch.float32) or st.accumulate
        padded = hd != hdp
        for B, sv in zip(reversed(self.blocks), reversed(cx['saved'])):
            x, m1, r1, y1, qkv, a, lses, x_mid, m2_, r2, y2, f1, g1, qkv_w, proj_w = sv
            d_g1 = B['fc2'].dx(dres)
            B['fc2'].dw(dres, g1)
            d_f1 = ops.act_bwd(f1, d_g1, ops.ACT_QUICK_GELU)
            d_y2 = B['fc1'].dx(d_f1)
            B['fc1'].dw(d_f1, y2)
            ops.layernorm_bwd(d_y2, x_mid, P[B['n2w']], m2_, r2, G.get(B['n2w']), G.get(B['n2b']), dx=dres, add_to_dx=True)
            d_a = ops.gemm(dres, proj_w, b_n=True)
            ops.colsum_(dres, G[B['proj_b']])
            if padded:
                tmp = ops.gemm(dres, a, a_t=True, b_n=True)                                     # [E, H*hdp]
                self._unpad_grad(B['proj_w'], tmp, (E, H, hdp), (E, H, hd), (slice(None), slice(None), slice(0, hd)))
            else:
                g = G[B['proj_w']]
                ops.gemm(dres, a, out=g, a_t=True, b_n=True, accumulate=acc(g))
            d_qkv = torch.zeros_like(qkv)
            for (o, nseq, L), lse in zip(cx['segs'], lses):
                v = slice(o, o + nseq * L)
                ops.attn_bwd(qkv[v, :H * hdp], qkv[v, H * hdp:2 * H * hdp], qkv[v, 2 * H * hdp:], a[v], d_a[v], lse,
                             d_qkv[v, :H * hdp], d_qkv[v, H * hdp:2 * H * hdp], d_qkv[v, 2 * H * hdp:], nseq, L, H, H, hdp, False, hd ** -0.5)
            ops.rope_(d_qkv, 0, 2 * H, hd, cx['rows'], cx['cos'], cx['sin'], inverse=True, head_stride=hdp, precise=True)
            d_y1 = ops.gemm(d_qkv, qkv_w, b_n=True)
            if padded:
                tmp = ops.gemm(d_qkv, y1, a_t=True, b_n=True)                                   # [3*H*hdp, E]
                self._unpad_grad(B['qkv_w'], tmp, (3, H, hdp, E), (3, H, hd, E), (slice(None), slice(None), slice(0, hd)))
                tb = torch.zeros(3 * H * hdp, dtype=torch.float32, device=dres.device)
                ops.colsum_(d_qkv, tb)
                G[B['qkv_b']].view(3, H, hd).add_(tb.view(3, H, hdp)[:, :, :hd])
            else:
                g = G[B['qkv_w']]
                ops.gemm(d_qkv, y1, out=g, a_t=True, b_n=True, accumulate=acc(g))
                ops.colsum_(d_qkv, G[B['qkv_b']])
            ops.layernorm_bwd(d_y1, x, P[B['n1w']], m1, r1, G.get(B['n1w']), G.get(B['n1b']), dx=dres, add_to_dx=True)
        g = G[self.patch_w]
        ops.gemm(dres, cx['pix'], out=g, a_t=True, b_n=True, accumulate=acc(g))


class NativeQwen2VL(NativeCausalLM):
    """hf:models/qwen2_vl/modeling_qwen2_vl.py:1207+ Qwen2VLForConditionalGeneration (align_anything/models/qwen2_vl.py):
    vision tower -> merged image features scattered over the image-token positions -> Qwen2 decoder (the Llama block with
    q/k/v biases, GQA) under multimodal RoPE.  `freeze_vision_tower` (default True here) covers the patch embedding and the
    visual blocks, the PatchMerger follows `freeze_mm_proj`.  The reference freezes by the substrings 'vision_tower' /
    'multi_modal_projector' (models/pretrained_model.py:265-281), which match nothing in `model.visual.*`: there the whole
    tower trains whatever the flags say -- pass freeze_vision_tower=False to reproduce that."""

    kind = 'qwen2vl'

    def __init__(self, cfg, device, trainable=True, freeze_mm_proj=False, freeze_language_model=False,
                 freeze_vision_tower=True, head='lm', dtype=bf16):
        super().__init__(cfg, device, trainable, dtype)
        self.head_kind = head
        t = cfg['text']
        self.hidden_size = t['hidden_size']
        self.train_lm = trainable and not freeze_language_model
        self.train_proj = trainable and not freeze_mm_proj
        self.train_tower = trainable and not freeze_vision_tower
        if self.train_tower and not self.train_proj:
            raise NotImplementedError('training the visual blocks with a frozen merger is not built')
        st = self.store
        self.vision = Qwen2VLVisionTower(cfg['vision'], st, 'model.visual.', self.train_proj, self.train_tower)
        self.embed = st.add('model.language_model.embed_tokens.weight', (t['vocab_size'], t['hidden_size']), self.train_lm, f32_grad=True)
        self.stack = LlamaStack(t, st, 'model.language_model.', self.train_lm)
        if head == 'lm':
            lm = st.add('lm_head.weight', (t['vocab_size'], t['hidden_size']), self.train_lm)
            self.head = LMHead(st, 'rms', self.stack.norm, None, lm, t['rms_eps'], self.train_lm)
        else:
            sw = st.add('score_head.weight', (1, t['hidden_size']), trainable, f32_grad=True)
            self.head = ScoreHead(st, 'rms', self.stack.norm, None, sw, t['rms_eps'], trainable)
        hd = t['head_dim']
        self.inv_freq = (1.0 / (t['rope_theta'] ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))).to(self.device)
        self._deltas = None
        self.finalize()

    def _padcols(self):
        return {'model.visual.patch_embed.proj.weight': True}

    def _unpad(self):
        v = self.cfg['vision']
        return {'model.visual.patch_embed.proj.weight':
                (v['embed_dim'], v['in_channels'], v['temporal_patch_size'], v['patch_size'], v['patch_size'])}

    def load_state_dict(self, sd, strict=True):
        self.vision.invalidate()
        return super().load_state_dict(sd, strict)

    def vision_features(self, pixel_values, image_grid_thw):
        return self.vision.forward(pixel_values, image_grid_thw)[0]

    def decode_start_positions(self, valid):
        return valid + self._deltas.to(valid.dtype) if self._deltas is not None else valid

    def forward_stream(self, input_ids, attention_mask=None, pixel_values=None, save=False, image_features=None,
                       position_ids=None, kv_sink=None, image_grid_thw=None, position_ids3=None, kv_len=None, pack=None):
        """position_ids3 (int32 [3, N, T], e.g. precomputed by the input pipeline on the host) avoids the device->host
        copy of input_ids that computing the 3-D rope index needs.  kv_len (int32 [N], optional): keys at or beyond it are masked
        in the decoder (right padding, as for NativeLlava: the reward model reads position -1, models/qwen2_vl.py:61-64).
        pack: shared-prompt packing plan (trainers.common.build_pack_plan).  The image tokens of a pair then appear ONCE in the packed ids, so the tower runs on
        the first half of the images only (the collator stacks them twice; the trainer checks it).  The 3-D rope index is still that of the [N, T] layout --
        it counts attended tokens, not slots, so both rows of a pair give their common prefix the same positions whatever their padding -- and a packed row
        is rotated with the table row of the slot that owns it."""
        N, T, Mp, start, _ = self._token_geometry(input_ids, attention_mask, None)
        self.stack.kv_len = kv_len
        P, t = self.store.p, self.cfg['text']
        ids = input_ids.reshape(-1)
        if pack is not None:
            if kv_sink is not None or kv_len is not None:
                raise RuntimeError('shared-prompt packing is a training-forward layout (no KV-cache prefill, no right padding)')
            ids = pack['ids']
        elif Mp != N * T:
            ids = torch.cat([ids, torch.zeros(Mp - N * T, dtype=ids.dtype, device=ids.device)])
        slot = feat = None
        has_img = pixel_values is not None or image_features is not None
        if has_img:
            if image_grid_thw is None:
                raise ValueError('Qwen2-VL needs image_grid_thw with pixel_values (the processor returns both)')
            if image_features is not None:
                feat, nf = image_features, None
            elif pack is not None:
                grid = image_grid_thw.tolist() if isinstance(image_grid_thw, torch.Tensor) else [list(g) for g in image_grid_thw]
                half = grid[:len(grid) // 2]
                feat, nf = self.vision.forward(pixel_values[:sum(a * b * c for a, b, c in half)], half, save=save and self.train_proj)
            else:
                feat, nf = self.vision.forward(pixel_values, image_grid_thw, save=save and self.train_proj)
            slot, count = ops.image_slot_index(ids, self.cfg['image_token_id'])
            self._last_image_token_count, self._last_feature_rows = count, nf
        if position_ids3 is None:
            if has_img:
                grid = image_grid_thw.tolist() if isinstance(image_grid_thw, torch.Tensor) else image_grid_thw
                p3, deltas = qwen2vl_rope_index(input_ids, attention_mask, grid, self.cfg['image_token_id'],
                                                self.cfg['vision']['spatial_merge_size'])
                position_ids3 = torch.from_numpy(p3).to(self.device)
                self._deltas = torch.from_numpy(deltas).to(self.device)
            else:   # text only: 1-D positions from the mask on all three axes (hf compute_3d_position_ids fallback)
                am = attention_mask.to(torch.int64) if attention_mask is not None else torch.ones_like(input_ids)
                p1 = ((torch.cumsum(am, 1) - 1) * am).to(torch.int32)
                position_ids3 = p1[None].expand(3, -1, -1)
                self._deltas = None
        p3 = position_ids3.to(torch.int32).reshape(3, N * T)
        if Mp != N * T:
            p3 = torch.cat([p3, torch.zeros((3, Mp - N * T), dtype=torch.int32, device=self.device)], 1)
        p3 = p3.contiguous()
        tables = ops.mrope_tables(p3, self.inv_freq, t['mrope_section'], self.dtype)
        rows = torch.arange(Mp, dtype=torch.int32, device=self.device)
        x = ops.embed_fwd(ids, P[self.embed], slot, feat)
        if save:
            self._ctx = dict(ids=ids, slot=slot, feat_rows=None if feat is None else feat.shape[0], N=N, T=T, start=start, pos=rows, pack=pack)
        if pack is not None:
            return self.stack.forward(x, N, T, start, pack['row2slot'].clamp(min=0), save, None, tables=tables, pack=pack)
        return self.stack.forward(x, N, T, start, rows, save, kv_sink, tables=tables)

    def embed_tokens(self, ids, pos=None):
        return ops.embed_fwd(ids, self.store.p[self.embed])

    def validate_batch(self):
        c = int(self._last_image_token_count.item())
        if self._last_feature_rows is not None and c != self._last_feature_rows:
            raise ValueError(f'Image features and image tokens do not match: tokens: {c}, features {self._last_feature_rows}')

    def backward_stream(self, dres, on_layer_done=None):
        cx = self._ctx
        dx = self.stack.backward(dres, cx['N'], cx['T'], cx['start'], cx['pos'], on_layer_done, pack=cx.get('pack'))
        G = self.store.g
        want_feat = cx['slot'] is not None and self.train_proj and self.vision._ctx is not None
        dfeat = torch.zeros((cx['feat_rows'], self.hidden_size), dtype=self.dtype, device=self.device) if want_feat else None
        if self.train_lm or want_feat:
            ops.embed_bwd(cx['ids'], dx, self.cfg['text']['vocab_size'], slot=cx['slot'],
                          dE=G.get(self.embed) if self.train_lm else None, dfeat=dfeat)
        if want_feat:
            self.vision.backward_merger(dfeat)


# ====================================================================== Qwen2-Audio
class Qwen2AudioTower:
    """hf:models/qwen2_audio/modeling_qwen2_audio.py:289-406 Qwen2AudioEncoder (the Whisper encoder + AvgPool1d(2)), forward AND
    backward -- the reference trains it by default (configs/train/text_audio_to_text/dpo.yaml:63).  Conv1d front-end = im2col +
    GEMM on the HF weights as stored; q/k/v fused (k_proj has no bias: its slice of the fused bias is held at zero); encoder
    attention masks the keys beyond each audio's length (`kv_len`); every activation is token-major [rows64(B*T), d]."""

    def __init__(self, acfg: dict, store: ParamStore, prefix: str, trainable: bool):
        self.cfg, self.store, self.prefix, self.trainable = acfg, store, prefix, trainable
        d, F, mel, S = acfg['d_model'], acfg['ffn_dim'], acfg['num_mel_bins'], acfg['max_source_positions']
        self.hd = d // acfg['num_heads']
        if self.hd not in (64, 128):
            raise NotImplementedError(f'audio head_dim {self.hd}: attention kernels are built for 64 and 128')
        if (3 * mel) % 64 or (3 * d) % 64:
            raise NotImplementedError('conv front-end: 3 * num_mel_bins and 3 * d_model must be multiples of 64')
        tr = trainable
        self.conv1 = Linear(store, store.add(prefix + 'conv1.weight', (d, 3 * mel), tr), store.add(prefix + 'conv1.bias', (d,), tr))
        self.conv2 = Linear(store, store.add(prefix + 'conv2.weight', (d, 3 * d), tr), store.add(prefix + 'conv2.bias', (d,), tr))
        self.pos = store.add(prefix + 'embed_positions.weight', (S, d), False)
        self.layers = []
        for i in range(acfg['num_layers']):
            p = f'{prefix}layers.{i}.'
            L = {'ln1w': store.add(p + 'self_attn_layer_norm.weight', (d,), tr), 'ln1b': store.add(p + 'self_attn_layer_norm.bias', (d,), tr)}
            blk = store.add_fused(p + 'self_attn.qkv_fused', [(p + 'self_attn.q_proj.weight', d, d), (p + 'self_attn.k_proj.weight', d, d),
                                                              (p + 'self_attn.v_proj.weight', d, d)], tr)
            bq = store.add(p + 'self_attn.qkv_fused.bias', (3 * d,), tr)
            del store.alias[bq]
            store.alias[p + 'self_attn.q_proj.bias'] = (bq, 0, (d,))
            store.alias[p + 'self_attn.v_proj.bias'] = (bq, 2 * d, (d,))
            L['qkv'], L['kbias'] = Linear(store, blk, bq), bq
            L['out'] = Linear(store, store.add(p + 'self_attn.out_proj.weight', (d, d), tr), store.add(p + 'self_attn.out_proj.bias', (d,), tr))
            L['ln2w'] = store.add(p + 'final_layer_norm.weight', (d,), tr)
            L['ln2b'] = store.add(p + 'final_layer_norm.bias', (d,), tr)
            L['fc1'] = Linear(store, store.add(p + 'fc1.weight', (F, d), tr), store.add(p + 'fc1.bias', (F,), tr))
            L['fc2'] = Linear(store, store.add(p + 'fc2.weight', (d, F), tr), store.add(p + 'fc2.bias', (d,), tr))
            self.layers.append(L)
        self.lnp_w = store.add(prefix + 'layer_norm.weight', (d,), tr)
        self.lnp_b = store.add(prefix + 'layer_norm.bias', (d,), tr)
        self._ctx = None

    def padcols(self):
        return {self.prefix + 'conv1.weight': True, self.prefix + 'conv2.weight': True}

    def unpad(self):
        c = self.cfg
        return {self.prefix + 'conv1.weight': (c['d_model'], c['num_mel_bins'], 3), self.prefix + 'conv2.weight': (c['d_model'], c['d_model'], 3)}

    def forward(self, input_features, audio_lengths, save=False):
        """input_features [B, mel, 2 * max_source_positions] (fp32 / bf16); audio_lengths int32 [B] (device) = frames after conv2 that
        carry audio.  Returns [rows64(B * S / 2), d] (frame j of audio b at row b * S/2 + j)."""
        c, P, dt = self.cfg, self.store.p, self.store.dtype
        B, mel, Tin = input_features.shape
        d, H, hd, S = c['d_model'], c['num_heads'], self.hd, c['max_source_positions']
        if mel != c['num_mel_bins'] or Tin != 2 * S:
            raise ValueError(f'Qwen2Audio expects mel features [B, {c["num_mel_bins"]}, {2 * S}], got {tuple(input_features.shape)}')
        col1, _ = ops.conv1d_im2col(input_features.contiguous(), B, mel, Tin, 1, True, dt)
        p1 = self.conv1.fwd(col1)
        a1 = ops.act_fwd(p1, ops.ACT_GELU)
        col2, _ = ops.conv1d_im2col(a1, B, d, Tin, 2, False, dt)
        p2 = self.conv2.fwd(col2)
        a2 = ops.act_fwd(p2, ops.ACT_GELU)
        M = a2.shape[0]
        dev = a2.device
        rows = torch.arange(M, device=dev)
        x = ops.embed_fwd(rows, a2, pos=(rows % S).to(torch.int32), P=P[self.pos])          # + embed_positions
        saved = []
        for L in self.layers:
            y1, m1, r1 = ops.layernorm_fwd(x, P[L['ln1w']], P[L['ln1b']], 1e-5)
            qkv = L['qkv'].fwd(y1)
            attn, lse = ops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, S, H, H, hd, False, hd ** -0.5, kv_len=audio_lengths,
                                     out=torch.zeros_like(x) if M != B * S else None)
            x_mid = L['out'].fwd(attn, residual=x)
            y2, m2, r2 = ops.layernorm_fwd(x_mid, P[L['ln2w']], P[L['ln2b']], 1e-5)
            f1 = L['fc1'].fwd(y2)
            g1 = ops.act_fwd(f1, ops.ACT_GELU)
            x_out = L['fc2'].fwd(g1, residual=x_mid)
            if save:
                saved.append((x, m1, r1, y1, qkv, attn, lse, x_mid, m2, r2, y2, f1, g1))
            x = x_out
        pooled = ops.avgpool2(x, B * S // 2)
        out, mp, rp = ops.layernorm_fwd(pooled, P[self.lnp_w], P[self.lnp_b], 1e-5)
        if save:
            self._ctx = dict(col1=col1, p1=p1, col2=col2, p2=p2, saved=saved, pooled=pooled, mp=mp, rp=rp, B=B, Tin=Tin, M=M,
                             lens=audio_lengths)
        return out

    def backward(self, dout):
        """dout [rows64(B*S/2), d] with zero pad rows; accumulates every tower gradient into store.g."""
        cx, c, P, G = self._ctx, self.cfg, self.store.p, self.store.g
        B, Tin, M = cx['B'], cx['Tin'], cx['M']
        d, H, hd, S = c['d_model'], c['num_heads'], self.hd, c['max_source_positions']
        d_pooled = ops.layernorm_bwd(dout, cx['pooled'], P[self.lnp_w], cx['mp'], cx['rp'], G.get(self.lnp_w), G.get(self.lnp_b))
        dres = ops.avgpool2(d_pooled, B * S // 2, backward=True)
        if dres.shape[0] != M:
            raise RuntimeError('Qwen2AudioTower.backward: row padding mismatch')
        for L, sv in zip(reversed(self.layers), reversed(cx['saved'])):
            x, m1, r1, y1, qkv, attn, lse, x_mid, m2, r2, y2, f1, g1 = sv
            d_g1 = L['fc2'].dx(dres)
            L['fc2'].dw(dres, g1)
            d_f1 = ops.act_bwd(f1, d_g1, ops.ACT_GELU)
            d_y2 = L['fc1'].dx(d_f1)
            L['fc1'].dw(d_f1, y2)
            ops.layernorm_bwd(d_y2, x_mid, P[L['ln2w']], m2, r2, G.get(L['ln2w']), G.get(L['ln2b']), dx=dres, add_to_dx=True)
            d_attn = L['out'].dx(dres)
            L['out'].dw(dres, attn)
            d_qkv = torch.zeros_like(qkv) if M != B * S else torch.empty_like(qkv)
            ops.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], attn, d_attn, lse, d_qkv[:, :d], d_qkv[:, d:2 * d], d_qkv[:, 2 * d:],
                         B, S, H, H, hd, False, hd ** -0.5, kv_len=cx['lens'])
            d_y1 = L['qkv'].dx(d_qkv)
            L['qkv'].dw(d_qkv, y1)
            if L['kbias'] in G:
                G[L['kbias']][d:2 * d].zero_()            # k_proj has no bias in HF: its slot of the fused bias never moves
            ops.layernorm_bwd(d_y1, x, P[L['ln1w']], m1, r1, G.get(L['ln1w']), G.get(L['ln1b']), dx=dres, add_to_dx=True)
        d_p2 = ops.act_bwd(cx['p2'], dres, ops.ACT_GELU)             # embed_positions is frozen: d(a2) = d(x)
        self.conv2.dw(d_p2, cx['col2'])
        d_col2 = self.conv2.dx(d_p2)
        d_a1 = ops.conv1d_col2im(d_col2, B, d, Tin, S, 2)
        d_p1 = ops.act_bwd(cx['p1'], d_a1, ops.ACT_GELU)
        self.conv1.dw(d_p1, cx['col1'])
        self._ctx = None


class NativeQwen2Audio(NativeCausalLM):
    """hf:models/qwen2_audio/modeling_qwen2_audio.py:776+ Qwen2AudioForConditionalGeneration (align_anything/models/
    qwen2_audio.py): audio tower -> Linear projector -> the first `output_length` frames of every audio scattered over the
    (processor-expanded) audio-token positions -> Qwen2 decoder with 1-D RoPE.  Freeze flags follow the reference's
    substrings: audio_tower / multi_modal_projector / language_model (models/pretrained_model.py:265-281)."""

    kind = 'qwen2audio'

    def __init__(self, cfg, device, trainable=True, freeze_mm_proj=False, freeze_language_model=False, freeze_audio_tower=False,
                 freeze_vision_tower=True, freeze_audio_proj=False, head='lm', dtype=bf16):
        super().__init__(cfg, device, trainable, dtype)
        self.head_kind = head
        t = cfg['text']
        self.hidden_size = t['hidden_size']
        self.train_lm = trainable and not freeze_language_model
        self.train_proj = trainable and not freeze_mm_proj
        self.train_tower = trainable and not freeze_audio_tower
        st = self.store
        self.tower = Qwen2AudioTower(cfg['audio'], st, 'model.audio_tower.', self.train_tower)
        ad = cfg['audio']['d_model']
        self.proj = Linear(st, st.add('model.multi_modal_projector.linear.weight', (t['hidden_size'], ad), self.train_proj),
                           st.add('model.multi_modal_projector.linear.bias', (t['hidden_size'],), self.train_proj))
        self.embed = st.add('model.language_model.embed_tokens.weight', (t['vocab_size'], t['hidden_size']), self.train_lm, f32_grad=True)
        self.stack = LlamaStack(t, st, 'model.language_model.', self.train_lm)
        if head == 'lm':
            lm = st.add('lm_head.weight', (t['vocab_size'], t['hidden_size']), self.train_lm)
            self.head = LMHead(st, 'rms', self.stack.norm, None, lm, t['rms_eps'], self.train_lm)
        else:
            sw = st.add('score_head.weight', (1, t['hidden_size']), trainable, f32_grad=True)
            self.head = ScoreHead(st, 'rms', self.stack.norm, None, sw, t['rms_eps'], trainable)
        self.finalize()

    def _padcols(self):
        return self.tower.padcols()

    def _unpad(self):
        return self.tower.unpad()

    def forward_stream(self, input_ids, attention_mask=None, pixel_values=None, save=False, image_features=None,
                       position_ids=None, kv_sink=None, input_features=None, feature_attention_mask=None, pack=None):
        """pack: shared-prompt packing plan (trainers.common.build_pack_plan).  The audio tokens of a pair then appear ONCE in the packed ids, so the tower runs on
        the first half of `input_features` only (the collator stacks the clips twice, datasets/text_audio_to_text/preference.py:178-229; the trainer checks it)."""
        N, T, Mp, start, pos = self._token_geometry(input_ids, attention_mask, position_ids)
        P = self.store.p
        ids = input_ids.reshape(-1)
        if pack is not None:
            if position_ids is not None or kv_sink is not None:
                raise RuntimeError('shared-prompt packing is a training-forward layout (no explicit position ids, no KV-cache prefill)')
            ids, pos = pack['ids'], pack['pos']
            if input_features is not None:
                input_features = input_features[:N // 2]
                feature_attention_mask = feature_attention_mask[:N // 2] if feature_attention_mask is not None else None
        elif Mp != N * T:
            ids = torch.cat([ids, torch.zeros(Mp - N * T, dtype=ids.dtype, device=ids.device)])
        slot = feat = None
        actx = None
        if input_features is not None:
            B = input_features.shape[0]
            flen = feature_attention_mask.sum(-1) if feature_attention_mask is not None else \
                torch.full((B,), input_features.shape[-1], device=input_features.device)
            alen = (flen - 1) // 2 + 1                                 # hf _get_feat_extract_output_lengths
            olen = ((alen - 2) // 2 + 1).tolist()                     # host read: the gather plan has data-dependent size
            S2 = self.cfg['audio']['max_source_positions'] // 2
            tower_out = self.tower.forward(input_features, alen.to(device=self.device, dtype=torch.int32), save=save and self.train_tower)
            proj_all = self.proj.fwd(tower_out)                        # [rows64(B*S/2), h]
            idx = torch.cat([torch.arange(int(n)) + b * S2 for b, n in enumerate(olen)]).to(self.device)
            n_feat = idx.numel()
            if n_feat % 64:
                idx_p = torch.cat([idx, torch.zeros(_pad64(n_feat) - n_feat, dtype=idx.dtype, device=self.device)])
            else:
                idx_p = idx
            feat = ops.embed_fwd(idx_p, proj_all)                      # row gather of the valid frames (pad rows: copies of row 0)
            slot, count = ops.image_slot_index(ids, self.cfg['audio_token_id'])
            self._last_image_token_count, self._last_feature_rows = count, n_feat
            actx = dict(idx=idx, n_feat=n_feat, tower_out=tower_out, rows_all=proj_all.shape[0])
        x = ops.embed_fwd(ids, P[self.embed], slot, feat)
        if save:
            self._ctx = dict(ids=ids, slot=slot, N=N, T=T, start=start, pos=pos if pack is None else pack['pos_full'], audio=actx, pack=pack)
        return self.stack.forward(x, N, T, start, pos, save, kv_sink, pack=pack)

    def embed_tokens(self, ids, pos=None):
        return ops.embed_fwd(ids, self.store.p[self.embed])

    def validate_batch(self):
        c = int(self._last_image_token_count.item())
        if c != self._last_feature_rows:
            raise ValueError(f'Audio features and audio tokens do not match, tokens: {c}, features: {self._last_feature_rows}')

    def backward_stream(self, dres, on_layer_done=None):
        cx = self._ctx
        dx = self.stack.backward(dres, cx['N'], cx['T'], cx['start'], cx['pos'], on_layer_done, pack=cx.get('pack'))
        G, a = self.store.g, cx['audio']
        want_feat = a is not None and (self.train_proj or self.train_tower)
        dfeat = torch.zeros((_pad64(a['n_feat']), self.hidden_size), dtype=self.dtype, device=self.device) if want_feat else None
        if self.train_lm or want_feat:
            ops.embed_bwd(cx['ids'], dx, self.cfg['text']['vocab_size'], slot=cx['slot'],
                          dE=G.get(self.embed) if self.train_lm else None, dfeat=dfeat)
        if want_feat:
            d_all = torch.zeros((a['rows_all'], self.hidden_size), dtype=self.dtype, device=self.device)
            d_all.index_copy_(0, a['idx'], dfeat[:a['n_feat']])      # rows of the frames that were scattered; everything else 0
            if self.train_proj:
                self.proj.dw(d_all, a['tower_out'])
            if self.train_tower:
                self.tower.backward(self.proj.dx(d_all))


# ====================================================================== Qwen3-MoE
class Qwen3MoeStack:
    """hf:models/qwen3_moe/modeling_qwen3_moe.py:305-350 x num_layers: attention with per-head RMSNorm on q and k (:143-166),
    sparse MoE block (:210-283).  Routing, the expert-major layout (`aa_moe_plan`), token movement and combine are HIP kernels
    (csrc/moe.hip); all experts of a layer run in ONE grouped-GEMM launch per matrix (`aa_gemm_grouped_*`: the kernel reads the
    128-row-tile -> expert table and the segment offsets from device memory), so a layer needs no host read at all."""

    def __init__(self, cfg: dict, store: ParamStore, prefix: str, trainable: bool, ep=None):
        """ep: expert_parallel.ExpertParallel -> this rank holds experts [e0, e0 + El) only (expert_parallel.py); None = all."""
        self.cfg, self.store, self.prefix, self.trainable, self.ep = cfg, store, prefix, trainable, ep
        h, F, E = cfg['hidden_size'], cfg['moe_intermediate_size'], cfg['num_experts']
        H, Hkv, hd = cfg['num_heads'], cfg['num_kv_heads'], cfg['head_dim']
        if hd not in (64, 128):
            raise NotImplementedError(f'head_dim {hd}: attention kernels are built for 64 and 128')
        tr = trainable
        e0, El = ep.local_experts(E) if ep is not None else (0, E)
        shard = (e0, E) if ep is not None else None
        self.layers = []
        for i in range(cfg['num_layers']):
            p = f'{prefix}layers.{i}.'
            L = {'ln1': store.add(p + 'input_layernorm.weight', (h,), tr),
                 # q | k | v stored as ONE block: one forward GEMM, one dX and one dW GEMM per layer on the big-tile kernel instead of three launches each on the
                 # small-tile one (15 ms of the 150 ms 12-layer step, profiles/r04_qwen3moe_kernel_stats.csv); the HF names stay aliases of its row ranges
                 'qkv': Linear(store, store.add_fused(p + 'self_attn.qkv_fused', [(p + 'self_attn.q_proj.weight', H * hd, h), (p + 'self_attn.k_proj.weight', Hkv * hd, h),
                                                                                 (p + 'self_attn.v_proj.weight', Hkv * hd, h)], tr)),
                 'q': Linear(store, p + 'self_attn.q_proj.weight'), 'k': Linear(store, p + 'self_attn.k_proj.weight'), 'v': Linear(store, p + 'self_attn.v_proj.weight'),
                 'qn': store.add(p + 'self_attn.q_norm.weight', (hd,), tr), 'kn': store.add(p + 'self_attn.k_norm.weight', (hd,), tr),
                 'o': Linear(store, store.add(p + 'self_attn.o_proj.weight', (h, H * hd), tr)),
                 'ln2': store.add(p + 'post_attention_layernorm.weight', (h,), tr),
                 'gate': Linear(store, store.add(p + 'mlp.gate.weight', (E, h), tr)),
                 'gu': store.add(p + 'mlp.experts.gate_up_proj', (El, 2 * F, h), tr, shard=shard),
                 'down': store.add(p + 'mlp.experts.down_proj', (El, h, F), tr, shard=shard)}
            self.layers.append(L)
        self.norm = store.add(prefix + 'norm.weight', (h,), tr)
        self.cos = self.sin = None
        self.saved = []
        self.tail, self.tail_used = None, False      # dead-row elimination in the last layer, as LlamaStack

    def _tables(self, T):
        if self.cos is None or self.cos.shape[0] < T:
            n = max(T, self.cfg.get('max_position_embeddings', 0) or T)
            self.cos, self.sin = rope_tables(n, self.cfg['head_dim'], self.cfg['rope_theta'], self.store.device, self.store.dtype, self.cfg.get('rope_scaling'))

    def kv_width(self):
        return 2 * self.cfg['num_kv_heads'] * self.cfg['head_dim']

    def _experts(self, L, n2, x_mid, rows):
        """Sparse MoE block without saved state (prefill of big batches / decode beyond a handful of rows): the training layout."""
        c, P = self.cfg, self.store.p
        logits = L['gate'].fwd(n2) if n2.shape[0] > 16 else ops.linear_small(n2, L['gate'].w)
        _, idx, w = ops.moe_route(logits, c['num_experts_per_tok'], c['norm_topk_prob'])
        if self.ep is not None:
            # expert-parallel weights: the routed rows travel to the ranks that own their experts and back, exactly as in the
            # training forward (one exchange per block; during a rollout that is one per decode position, with every rank
            # stepping in lockstep -- generation.py keeps the ranks' step counts equal)
            return self._ep_experts_fwd(L, n2, x_mid, idx, w, rows)[0]
        plan = ops.moe_plan(idx, c['num_experts'])
        gu = ops.gemm_grouped(ops.moe_gather(n2, plan['src']), P[L['gu']], plan)
        yp = ops.gemm_grouped(ops.swiglu_fwd(gu), P[L['down']], plan)
        return ops.moe_combine(yp, plan['pos'], w, rows, residual=x_mid)

    def decode_step(self, x, cache, t, Tmax, pos, start, length):
        """One new token per sequence against the KV cache (generation.py).  The experts of a handful of rows are streamed once
        per routed (token, choice) row by `aa_moe_gemv_bf16`; beyond that the 128-row-tile training layout is cheaper (every
        expert matrix is read once however many rows chose it)."""
        c, P = self.cfg, self.store.p
        H, Hkv, hd, eps, E, k = c['num_heads'], c['num_kv_heads'], c['head_dim'], c['rms_eps'], c['num_experts'], c['num_experts_per_tok']
        qw, kw = H * hd, Hkv * hd
        N = x.shape[0]
        self._tables(Tmax)
        rows = torch.arange(N, device=x.device)
        ident = torch.arange(N * k, dtype=torch.int32, device=x.device).view(N, k)
        for li, L in enumerate(self.layers):
            n1, _ = ops.rmsnorm_fwd(x, P[L['ln1']], eps)
            q, kk, v = ops.linear_small(n1, L['q'].w), ops.linear_small(n1, L['k'].w), ops.linear_small(n1, L['v'].w)
            qn, _ = ops.rmsnorm_fwd(q.view(N * H, hd), P[L['qn']], eps)
            kn, _ = ops.rmsnorm_fwd(kk.view(N * Hkv, hd), P[L['kn']], eps)
            qn, kn = qn.view(N, qw), kn.view(N, kw)
            ops.rope_(qn, 0, H, hd, pos, self.cos, self.sin)
            ops.rope_(kn, 0, Hkv, hd, pos, self.cos, self.sin)
            cl = cache[li]
            cl.view(N, Tmax, 2 * kw).index_put_((rows, t), torch.cat([kn, v], dim=1))
            attn = ops.attn_decode(qn, cl, cl[:, kw:], Tmax, start, length, N, H, Hkv, hd, hd ** -0.5)
            x_mid = ops.linear_small(attn, L['o'].w, residual=x)
            n2, _ = ops.rmsnorm_fwd(x_mid, P[L['ln2']], eps)
            if self.ep is not None or N > 16 or N * k > 2 * E:      # expert-parallel: the experts are not all here, rows travel
                x = self._experts(L, n2, x_mid, N)
                continue
            _, idx, w = ops.moe_route(ops.linear_small(n2, L['gate'].w), k, c['norm_topk_prob'])
            gu = ops.moe_gemv(n2, P[L['gu']], idx, k)                  # [N*k, 2F]: row (token, choice) x its expert
            yp = ops.moe_gemv(ops.swiglu_fwd(gu), P[L['down']], idx, 1)
            x = ops.moe_combine(yp, ident, w, N, residual=x_mid)
        return x

    # ---- the experts this rank holds, on rows already in the 128-row-tile expert-major layout `plan`
    def _local_experts_fwd(self, L, xp, plan):
        P = self.store.p
        gu = ops.gemm_grouped(xp, P[L['gu']], plan)                # every tile is written (zeros where a tile has no expert): ops.grouped_out
        act = ops.swiglu_fwd(gu)
        yp = ops.gemm_grouped(act, P[L['down']], plan)
        return gu, act, yp

    def _local_experts_bwd(self, L, dyp, plan, xp, gu, act):
        """dyp: gradient of the expert outputs in the tile layout (pad rows zero) -> gradient of xp; dW into the store."""
        P, G, st, tr = self.store.p, self.store.g, self.store, self.trainable
        acc = lambda g: (g.dtype == torch.float32) or st.accumulate
        dact = ops.gemm_grouped(dyp, P[L['down']], plan, b_n=True)
        if tr:      # experts that saw no token get an all-zero gradient from the kernel (never a stale one)
            ops.gemm_grouped_dw(dyp, act, plan, G[L['down']], accumulate=acc(G[L['down']]))
        dgu = ops.swiglu_bwd(gu, dact)
        dxp = ops.gemm_grouped(dgu, P[L['gu']], plan, b_n=True)
        if tr:
            ops.gemm_grouped_dw(dgu, xp, plan, G[L['gu']], accumulate=acc(G[L['gu']]))
        return dxp

    # ---- expert parallelism (expert_parallel.py): rows travel to the ranks that own their experts and back
    def _ep_experts_fwd(self, L, n2, x_mid, idx, w, Mp):
        ep, E = self.ep, self.cfg['num_experts']
        lay = ops.moe_plan(idx, E, align=1)                      # dense expert-major order = send order (experts are rank-contiguous)
        if ep.padded:
            # sync-free form (expert_parallel.py): constant-size blocks per peer, counts and the local plan stay on the device
            cap = ep.capacity(ep.shared_pairs(idx.numel()))     # the same block size on every rank, whatever their padded batch shapes
            send_src, pos_p = ep.padded_send_layout(lay['counts'], lay['src'], lay['pos'], idx, cap)
            xr = ep.exchange_fixed(ops.moe_gather(n2, send_src))                # [size * cap, h]: block s = rank s's rows for my experts, zero tail
            ids = ep.padded_recv_ids(ep.exchange_counts_device(lay['counts']), cap)
            plan = ops.moe_plan(ids, E // ep.size, allow_invalid=True)
            xp = ops.moe_gather(xr, plan['src'])
            gu, act, yp = self._local_experts_fwd(L, xp, plan)
            ys = ep.exchange_fixed(ops.moe_combine(yp, plan['pos'], None, xr.shape[0]))      # back in my padded send order
            x_out = ops.moe_combine(ys, pos_p, w, Mp, residual=x_mid)
            return x_out, {'lay': lay, 'local': plan, 'pos_p': pos_p, 'ys': ys}, xp, gu, act, yp
        xs = ops.moe_gather(n2, lay['src'])                      # [Mp*k, h]
        send, recv, recv_counts = ep.exchange_counts(lay['counts'])
        xr = ep.exchange_rows(xs, send, recv)                    # rows for my experts, source-rank major
        plan = ops.moe_plan(ep.local_expert_ids(recv_counts, n2.device), E // ep.size)
        xp = ops.moe_gather(xr, plan['src'])
        gu, act, yp = self._local_experts_fwd(L, xp, plan)
        yr = ops.moe_combine(yp, plan['pos'], None, xr.shape[0])         # tile layout -> arrival order (a copy: k = 1, unit weight)
        ys = ep.exchange_rows(yr, recv, send)                    # back in my send order
        x_out = ops.moe_combine(ys, lay['pos'], w, Mp, residual=x_mid)
        return x_out, {'lay': lay, 'local': plan, 'send': send, 'recv': recv, 'ys': ys}, xp, gu, act, yp

    def _ep_experts_bwd(self, L, dres, ctx, xp, gu, act, w, Mp):
        ep, lay, plan = self.ep, ctx['lay'], ctx['local']
        if 'pos_p' in ctx:                                       # the same fixed-size exchanges with gradients
            dys, dw = ops.moe_combine_bwd(dres, ctx['ys'], ctx['pos_p'], w)
            dyr = ep.exchange_fixed(dys)
            dxp = self._local_experts_bwd(L, ops.moe_gather(dyr, plan['src']), plan, xp, gu, act)
            dxs = ep.exchange_fixed(ops.moe_combine(dxp, plan['pos'], None, dyr.shape[0]))
            return ops.moe_combine(dxs, ctx['pos_p'], None, Mp), dw
        dys, dw = ops.moe_combine_bwd(dres, ctx['ys'], lay['pos'], w)
        dyr = ep.exchange_rows(dys, ctx['send'], ctx['recv'])
        dxp = self._local_experts_bwd(L, ops.moe_gather(dyr, plan['src']), plan, xp, gu, act)
        dxr = ops.moe_combine(dxp, plan['pos'], None, dyr.shape[0])
        dxs = ep.exchange_rows(dxr, ctx['recv'], ctx['send'])
        return ops.moe_combine(dxs, lay['pos'], None, Mp), dw

    def forward(self, x, N, T, start, pos, save, kv_sink=None, pack=None):
        """pack (trainers.common.build_pack_plan; shared-prompt packing, as LlamaStack.forward): x / pos are PACKED token rows -- norms, projections, per-head
        norms + rotary embedding, the router and the experts see a pair's common prefix once (its routing is that of either copy: same hidden state, same
        top-k) -- and attention runs on the [N, T] layout between row gathers."""
        c, P = self.cfg, self.store.p
        H, Hkv, hd, eps, E, k = c['num_heads'], c['num_kv_heads'], c['head_dim'], c['rms_eps'], c['num_experts'], c['num_experts_per_tok']
        self._tables(T)
        self.saved = []
        Mp = x.shape[0]
        if pack is not None and (self.ep is not None or kv_sink is not None):
            raise RuntimeError('shared-prompt packing of the MoE stack: single-rank experts, training forward only')
        # `tail` (LlamaStack.forward): the last layer's attention queries, o-projection, router and experts on the response-window rows only
        tail, self.tail = self.tail, None
        if tail is not None and (kv_sink is not None or self.ep is not None or not self.layers):
            tail = None
        self._tail_saved = tail if save else None
        self.tail_used = tail is not None
        if self.ep is not None and self.ep.padded and self.ep._scope is None:
            # every MoE block of this forward exchanges Mp x k pairs: the ranks agree on the block size once (expert_parallel.pass_scope)
            with self.ep.pass_scope(Mp * k):
                return self.forward(x, N, T, start, pos, save, kv_sink)
        for li, L in enumerate(self.layers):
            n1, rstd1 = ops.rmsnorm_fwd(x, P[L['ln1']], eps)
            # one projection GEMM; q / k / v are column slices of its output.  The per-head norm + rotary embedding read their slice in place and
            # write dense rotated q / k (aa_rmsnorm_rope_fwd: one pass each, same bits as aa_rmsnorm_fwd + aa_rope_inplace); attention reads v in place
            qkv = L['qkv'].fwd(n1)
            q, kk, v = qkv[:, :H * hd], qkv[:, H * hd:(H + Hkv) * hd], qkv[:, (H + Hkv) * hd:]
            qn, rq = ops.rmsnorm_rope_fwd(q, P[L['qn']], eps, pos, self.cos, self.sin, H, hd=hd)
            kn, rk = ops.rmsnorm_rope_fwd(kk, P[L['kn']], eps, pos, self.cos, self.sin, Hkv, hd=hd)
            qn, kn = qn.view(Mp, H * hd), kn.view(Mp, Hkv * hd)
            if kv_sink is not None:     # post-norm, post-RoPE keys | values of this layer -> KV cache (prefill)
                kv_sink(li, torch.cat([kn[:N * T], v[:N * T]], dim=1))
            if pack is not None:
                qf = ops.moe_gather(qn, pack['slot2row'])                      # the [N, T] layout (pad slots: zero rows), kept for the backward
                kvf = ops.moe_gather(torch.cat([kn, v], dim=1), pack['slot2row'])
                kf, vf = kvf[:, :Hkv * hd], kvf[:, Hkv * hd:]
                Mf = qf.shape[0]
                last = tail is not None and li == len(self.layers) - 1
                attn_full, lse = ops.attn_fwd(qf, kf, vf, N, T, H, Hkv, hd, True, hd ** -0.5, start,
                                              out=None if Mf == N * T else torch.zeros((Mf, H * hd), dtype=x.dtype, device=x.device),
                                              q_skip=tail['qskip'] if last else pack.get('qskip'), work_frac=tail['frac'] if last else pack.get('attn_frac', 1.0))
                attn = ops.embed_fwd(tail['gather_attn'], attn_full) if last else ops.moe_gather(attn_full, pack['row2slot'])
                qn, kn, v = qf, kf, vf
            else:
                last = tail is not None and li == len(self.layers) - 1
                attn, lse = ops.attn_fwd(qn, kn, v, N, T, H, Hkv, hd, True, hd ** -0.5, start,
                                         out=None if Mp == N * T else torch.zeros((Mp, H * hd), dtype=x.dtype, device=x.device),
                                         q_skip=tail['qskip'] if last else None, work_frac=tail['frac'] if last else 1.0)
                attn_full = attn
                if last:
                    attn = ops.embed_fwd(tail['gather_attn'], attn_full)
            x_in = x
            if last:                                                           # from here on this layer lives on the window rows
                x = ops.embed_fwd(tail['gather_x'], x)
                Mp = x.shape[0]
            x_mid = L['o'].fwd(attn, residual=x)
            n2, rstd2 = ops.rmsnorm_fwd(x_mid, P[L['ln2']], eps)
            logits = L['gate'].fwd(n2)
            probs, idx, w = ops.moe_route(logits, k, c['norm_topk_prob'])
            if self.ep is not None:
                x_out, plan, xp, gu, act, yp = self._ep_experts_fwd(L, n2, x_mid, idx, w, Mp)
            else:
                plan = ops.moe_plan(idx, E)
                xp = ops.moe_gather(n2, plan['src'])                           # [cap, h], zero pad rows
                gu, act, yp = self._local_experts_fwd(L, xp, plan)
                x_out = ops.moe_combine(yp, plan['pos'], w, Mp, residual=x_mid)
            if save:
                self.saved.append((x_in, rstd1, n1, q, kk, v, rq, rk, qn, kn, attn, lse, x_mid, rstd2, n2, probs, idx, w, plan, xp, gu, act, yp, attn_full))
            x = x_out
        return x

    def backward(self, dres, N, T, start, pos, on_layer_done=None, pack=None):
        c, P, G = self.cfg, self.store.p, self.store.g
        H, Hkv, hd, E = c['num_heads'], c['num_kv_heads'], c['head_dim'], c['num_experts']
        tr, st = self.trainable, self.store
        Mp = dres.shape[0]
        tail, self._tail_saved = getattr(self, '_tail_saved', None), None        # set: `dres` arrives as [rows_pad, h] in window order (see forward)
        for L, sv in zip(reversed(self.layers), reversed(self.saved)):
            x, rstd1, n1, q, kk, v, rq, rk, qn, kn, attn, lse, x_mid, rstd2, n2, probs, idx, w, plan, xp, gu, act, yp, attn_full = sv
            sv = None
            # ---- sparse MoE block
            if 'lay' in plan:
                d_n2, dw = self._ep_experts_bwd(L, dres, plan, xp, gu, act, w, Mp)
            else:
                dyp, dw = ops.moe_combine_bwd(dres, yp, plan['pos'], w, src=plan['src'])
                d_n2 = ops.moe_combine(self._local_experts_bwd(L, dyp, plan, xp, gu, act), plan['pos'], None, Mp)
            dlogits = ops.moe_route_bwd(probs, idx, dw, c['norm_topk_prob'], x.dtype)
            if E % 64:   # the expert count is the contraction dim here: zero-pad it for small (test-size) routers
                Ep = _pad64(E)
                dl = torch.zeros((Mp, Ep), dtype=x.dtype, device=x.device); dl[:, :E] = dlogits
                wg = torch.zeros((Ep, c['hidden_size']), dtype=x.dtype, device=x.device); wg[:E] = L['gate'].w
                ops.gemm(dl, wg, out=d_n2, b_n=True, accumulate=True)
            else:
                ops.gemm(dlogits, L['gate'].w, out=d_n2, b_n=True, accumulate=True)
            if tr:
                L['gate'].dw(dlogits, n2)
            ops.rmsnorm_bwd(d_n2, x_mid, P[L['ln2']], rstd2, G.get(L['ln2']) if tr else None, dx=dres, add_to_dx=True)
            # ---- attention
            d_attn = L['o'].dx(dres)
            if tr:
                L['o'].dw(dres, attn)
            if tail is not None:                                              # back to the stack's rows / the [N, T] layout: zero gradient outside the windows
                d_attn = ops.moe_gather(d_attn, tail['scatter_attn'])
                dres = ops.moe_gather(dres, tail['scatter_x'])
                Mp = dres.shape[0]
            Mf = qn.shape[0]                                                  # rows of the [N, T] layout (== Mp unless packed)
            z = lambda t: torch.zeros_like(t) if Mf != N * T else torch.empty_like(t)
            # the gradient of the fused projection output: attention writes dV into its slice, the per-head norm backward dq / dk into theirs
            d_qkv = torch.zeros((Mp, (H + 2 * Hkv) * hd), dtype=qn.dtype, device=qn.device) if (Mp != N * T and pack is None) else torch.empty((Mp, (H + 2 * Hkv) * hd), dtype=qn.dtype, device=qn.device)
            if pack is not None:
                if tail is None:
                    d_attn = ops.moe_gather(d_attn, pack['owner'])            # the rejected copy of a shared prefix row is nobody's output
                dqn = z(qn)
                dkv = torch.zeros((Mf, 2 * Hkv * hd), dtype=qn.dtype, device=qn.device) if Mf != N * T else torch.empty((Mf, 2 * Hkv * hd), dtype=qn.dtype, device=qn.device)
                dkn, dv = dkv[:, :Hkv * hd], dkv[:, Hkv * hd:]
            else:
                dqn, dkn = z(qn), z(kn)
                dv = d_qkv[:, (H + Hkv) * hd:]
            fuse_rope = dqn.dtype == bf16 and ops.attn_rope_fused()         # as LlamaStack.backward
            ops.attn_bwd(qn, kn, v, attn_full, d_attn, lse, dqn, dkn, dv, N, T, H, Hkv, hd, True, hd ** -0.5, start,
                         rope=(pos, self.cos, self.sin) if fuse_rope else None,
                         q_skip=tail['qskip'] if tail is not None else (pack.get('qskip') if pack is not None else None),
                         work_frac=tail['frac'] if tail is not None else (pack.get('attn_frac', 1.0) if pack is not None else 1.0))
            tail = None                                                       # the last layer only
            if not fuse_rope:
                ops.rope_(dqn, 0, H, hd, pos, self.cos, self.sin, inverse=True)
                ops.rope_(dkn, 0, Hkv, hd, pos, self.cos, self.sin, inverse=True)
            if pack is not None:                                              # a shared row's gradient = the sum over its two copies
                dqn = ops.gather2_add(dqn, pack['row2slot'], pack['row2slot_b'])
                dkv = ops.gather2_add(dkv, pack['row2slot'], pack['row2slot_b'])
                dkn = dkv[:, :Hkv * hd].contiguous()
                d_qkv[:, (H + Hkv) * hd:] = dkv[:, Hkv * hd:]
            ops.rmsnorm_heads_bwd(dqn.view(Mp * H, hd), q, P[L['qn']], rq, G.get(L['qn']) if tr else None, d_qkv[:, :H * hd], H, hd)
            ops.rmsnorm_heads_bwd(dkn.view(Mp * Hkv, hd), kk, P[L['kn']], rk, G.get(L['kn']) if tr else None, d_qkv[:, H * hd:(H + Hkv) * hd], Hkv, hd)
            d_n1 = L['qkv'].dx(d_qkv)
            if tr:
                L['qkv'].dw(d_qkv, n1)
            ops.rmsnorm_bwd(d_n1, x, P[L['ln1']], rstd1, G.get(L['ln1']) if tr else None, dx=dres, add_to_dx=True)
            if on_layer_done is not None:
                on_layer_done(L)
        self.saved = []
        return dres


class NativeQwen3Moe(NativeCausalLM):
    """hf:models/qwen3_moe/modeling_qwen3_moe.py:597+ Qwen3MoeForCausalLM (align_anything/models/qwen3_moe.py), text only."""

    kind = 'qwen3moe'

    def __init__(self, cfg, device, trainable=True, head='lm', dtype=bf16, ep=None):
        super().__init__(cfg, device, trainable, dtype)
        self.head_kind = head
        self.hidden_size = cfg['hidden_size']
        self.ep = ep
        st = self.store
        self.embed = st.add('model.embed_tokens.weight', (cfg['vocab_size'], cfg['hidden_size']), trainable, f32_grad=True)
        self.stack = Qwen3MoeStack(cfg, st, 'model.', trainable, ep)
        if head == 'lm':
            lm = st.add('lm_head.weight', (cfg['vocab_size'], cfg['hidden_size']), trainable)
            self.head = LMHead(st, 'rms', self.stack.norm, None, lm, cfg['rms_eps'], trainable)
        else:   # models/qwen3_moe.py:36-72 AccustomedQwen3MoeRewardModel
            sw = st.add('score_head.weight', (1, cfg['hidden_size']), trainable, f32_grad=True)
            self.head = ScoreHead(st, 'rms', self.stack.norm, None, sw, cfg['rms_eps'], trainable)
        self.finalize()

    def forward_stream(self, input_ids, attention_mask=None, pixel_values=None, save=False, image_features=None,
                       position_ids=None, kv_sink=None, pack=None):
        N, T, Mp, start, pos = self._token_geometry(input_ids, attention_mask, position_ids)
        ids = input_ids.reshape(-1)
        if pack is not None:
            if position_ids is not None or kv_sink is not None:
                raise RuntimeError('shared-prompt packing is a training-forward layout (no explicit position ids, no KV-cache prefill)')
            ids, pos = pack['ids'], pack['pos']
        elif Mp != N * T:
            ids = torch.cat([ids, torch.zeros(Mp - N * T, dtype=ids.dtype, device=ids.device)])
        x = ops.embed_fwd(ids, self.store.p[self.embed])
        if save:
            self._ctx = dict(ids=ids, N=N, T=T, start=start, pos=pos if pack is None else pack['pos_full'], pack=pack)
        return self.stack.forward(x, N, T, start, pos, save, kv_sink, pack=pack)

    def embed_tokens(self, ids, pos=None):
        return ops.embed_fwd(ids, self.store.p[self.embed])

    def load_state_dict(self, sd, strict=True):
        """Accepts the fused expert tensors of transformers >= 5 (`mlp.experts.gate_up_proj` [E, 2F, h], `down_proj` [E, h, F]) and
        the per-expert keys hub checkpoints are written with (`mlp.experts.<e>.{gate,up,down}_proj.weight`), which are merged here
        exactly as hf:conversion_mapping.py "qwen2_moe" does (experts stacked on dim 0, gate before up).  Expert-parallel ranks
        merge only the experts they hold."""
        return super().load_state_dict(self.fuse_expert_keys(sd), strict)

    def fuse_expert_keys(self, sd):
        if not any('.mlp.experts.0.' in k for k in sd):
            return sd
        E = self.cfg['num_experts']
        e0, n = self.ep.local_experts(E) if self.ep is not None else (0, E)
        keep = lambda k: '.mlp.experts.' not in k or k.endswith(('experts.gate_up_proj', 'experts.down_proj'))
        gu = lambda p: (lambda: torch.stack([torch.cat([sd[f'{p}{e}.gate_proj.weight'], sd[f'{p}{e}.up_proj.weight']], 0) for e in range(e0, e0 + n)]))
        dn = lambda p: (lambda: torch.stack([sd[f'{p}{e}.down_proj.weight'] for e in range(e0, e0 + n)]))
        thunks = {}
        for i in range(self.cfg['num_layers']):
            p = f'model.layers.{i}.mlp.experts.'
            if p + '0.gate_proj.weight' in sd:
                thunks[p + 'gate_up_proj'], thunks[p + 'down_proj'] = gu(p), dn(p)
        if not isinstance(sd, dict):      # a lazy checkpoint (checkpoint.LazyCheckpoint): one layer's experts are merged when the store asks for them
            from .checkpoint import Derived
            return Derived(sd, lambda k: not keep(k), thunks)
        out = {k: v for k, v in sd.items() if keep(k)}
        out.update({k: f() for k, f in thunks.items()})
        return out

    def state_dict(self):
        """HF-layout tensors.  With expert parallelism this is a COLLECTIVE call: every rank contributes its expert rows and
        gets the full [E, ...] tensors back (what save_pretrained writes)."""
        sd = super().state_dict()
        if self.ep is not None:
            for name in self.store.shard:
                sd[name] = self.ep.all_gather_rows(sd[name])
        return sd

    def backward_stream(self, dres, on_layer_done=None):
        cx = self._ctx
        dx = self.stack.backward(dres, cx['N'], cx['T'], cx['start'], cx['pos'], on_layer_done, pack=cx.get('pack'))
        if self.trainable:
            ops.embed_bwd(cx['ids'], dx, self.cfg['vocab_size'], dE=self.store.g.get(self.embed))


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
        # config.tie_word_embeddings (Llama-3.2-1B / 3B, Qwen2.5-0.5B ... 3B): lm_head IS the embedding matrix -- one parameter, whose fp32 gradient
        # buffer takes the head's dW (accumulated) and the embedding's scatter-add, as NativeOPT does it
        self.tied = bool(cfg.get('tie_word_embeddings')) and head == 'lm'
        if head == 'lm':
            lm = self.embed if self.tied else st.add('lm_head.weight', (cfg['vocab_size'], cfg['hidden_size']), trainable)
            self.head = LMHead(st, 'rms', self.stack.norm, None, lm, cfg['rms_eps'], trainable)
        else:
            sw = st.add('score_head.weight', (1, cfg['hidden_size']), trainable, f32_grad=True)
            self.head = ScoreHead(st, 'rms', self.stack.norm, None, sw, cfg['rms_eps'], trainable)
        self.finalize()

    def load_state_dict(self, sd, strict=True):
        if self.tied and isinstance(sd, dict):
            sd = dict(sd)
            sd.pop('lm_head.weight', None)      # tied: the store has no tensor of that name (a lazy checkpoint's copy, if any, is never read)
        return super().load_state_dict(sd, strict)

    def state_dict(self):
        sd = super().state_dict()
        if self.tied:
            sd['lm_head.weight'] = sd['model.embed_tokens.weight']
        return sd

    def forward_stream(self, input_ids, attention_mask=None, pixel_values=None, save=False, image_features=None,
                       position_ids=None, kv_sink=None, pack=None):
        """pack: shared-prompt packing plan (trainers.common.build_pack_plan), as in NativeLlava.forward_stream."""
        N, T, Mp, start, pos = self._token_geometry(input_ids, attention_mask, position_ids)
        ids = input_ids.reshape(-1)
        if pack is not None:
            if position_ids is not None or kv_sink is not None:
                raise RuntimeError('shared-prompt packing is a training-forward layout (no explicit position ids, no KV-cache prefill)')
            ids, pos = pack['ids'], pack['pos']
        elif Mp != N * T:
            ids = torch.cat([ids, torch.zeros(Mp - N * T, dtype=ids.dtype, device=ids.device)])
        x = ops.embed_fwd(ids, self.store.p[self.embed])
        if save:
            self._ctx = dict(ids=ids, N=N, T=T, start=start, pos=pos if pack is None else pack['pos_full'], pack=pack)
        return self.stack.forward(x, N, T, start, pos, save, kv_sink, pack=pack)

    def embed_tokens(self, ids, pos=None):
        return ops.embed_fwd(ids, self.store.p[self.embed])

    def backward_stream(self, dres, on_layer_done=None):
        cx = self._ctx
        dx = self.stack.backward(dres, cx['N'], cx['T'], cx['start'], cx['pos'], on_layer_done, pack=cx.get('pack'))
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
        if isinstance(sd, dict):
            sd = dict(sd)
            sd.pop('lm_head.weight', None)  # tied (the store has no alias of that name: a lazy checkpoint's copy is simply never read)
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
    if cfg['kind'] == 'qwen2vl':
        return NativeQwen2VL(cfg, device, trainable, head=head, dtype=dtype, **freeze)
    if cfg['kind'] == 'qwen2audio':
        return NativeQwen2Audio(cfg, device, trainable, head=head, dtype=dtype, **freeze)
    if cfg['kind'] == 'qwen3moe':
        return NativeQwen3Moe(cfg, device, trainable, head=head, dtype=dtype, **freeze)      # ep=ExpertParallel(...)
    raise ValueError(f"no native model for kind {cfg['kind']!r}")
