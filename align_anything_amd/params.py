"""Flat parameter storage for the native engine.

MI355X-first memory layout: with 288 GB of HBM per GPU a 7B policy + fp32 master/m/v + grads + the frozen
reference fit on ONE device, so instead of DeepSpeed's ZeRO partitions (configs/deepspeed/ds_z3_config.json)
every parameter of an optimizer group lives in one flat bf16 buffer with matching flat fp32 master / m / v
buffers and one flat gradient buffer.  The optimizer step is then a few streaming kernels
(csrc/optim.hip) and the data-parallel gradient exchange is a few large RCCL all-reduces.

Groups (storage x weight-decay, following align_anything/utils/tools.py:241-270):
  mat    : matrices, bf16 gradients (GEMM epilogue output), weight decay
  emb    : tensors whose gradient is accumulated with fp32 atomics / fp32 GEMM epilogue, weight decay
  vec    : biases and norm weights (the reference's no-decay name list), fp32 gradients, no decay
  frozen : no gradient, no optimizer state
"""
from __future__ import annotations

from collections import OrderedDict

import torch

NO_DECAY = ('bias', 'layer_norm.weight', 'layernorm.weight', 'norm.weight', 'ln_f.weight')
ALIGN = 64  # elements; keeps every view 128-byte aligned


def is_no_decay(name: str) -> bool:
    return any(nd in name for nd in NO_DECAY)


class ParamStore:
    def __init__(self, device, dtype=torch.bfloat16):
        self.device = torch.device(device)
        # bfloat16 = production layout (bf16 weights + fp32 masters); float32 = parity mode: the model computes on the
        # fp32 masters themselves (flat[g] IS master[g]), every gradient is fp32
        if dtype not in (torch.bfloat16, torch.float32):
            raise ValueError(f'ParamStore dtype must be bfloat16 or float32, got {dtype}')
        self.dtype = dtype
        self.specs: OrderedDict[str, dict] = OrderedDict()   # storage blocks
        self.alias: dict[str, tuple[str, int, tuple]] = {}    # hf name -> (block, row offset, shape)
        self.flat: dict[str, torch.Tensor] = {}
        self.gflat: dict[str, torch.Tensor] = {}
        self.master: dict[str, torch.Tensor] = {}
        self.m: dict[str, torch.Tensor] = {}
        self.v: dict[str, torch.Tensor] = {}
        self.p: dict[str, torch.Tensor] = {}
        self.g: dict[str, torch.Tensor] = {}
        self.sizes: dict[str, int] = {}
        self.shard: dict[str, tuple[int, int]] = {}           # expert-parallel blocks: hf name -> (first row held here, full dim 0)
        self.accumulate = False   # set by NativeEngine during gradient accumulation (bf16 dW GEMMs then add)

    # ---- registration
    def add(self, name, shape, trainable=True, f32_grad=False, shard=None):
        """shard = (first, full): this rank holds rows [first, first + shape[0]) of a [full, ...] checkpoint tensor (expert
        parallelism).  Trainable shards live in their own flat group 'exp': its gradients are complete on the owning rank and
        are never all-reduced (engine.py)."""
        group = 'frozen'
        if shard is not None:
            self.shard[name] = (int(shard[0]), int(shard[1]))
        if trainable and shard is not None:
            group = 'exp'
        elif trainable:
            if is_no_decay(name):
                group = 'vec'
            elif f32_grad or len(shape) == 1:
                group = 'emb'
            else:
                group = 'mat'
        self.specs[name] = {'shape': tuple(shape), 'group': group}
        self.alias[name] = (name, 0, tuple(shape))
        return name

    def add_fused(self, block, parts, trainable=True):
        """parts: [(hf_name, rows)], all [rows_i, cols]; stored as one [sum rows, cols] block."""
        cols = parts[0][2]
        rows = sum(p[1] for p in parts)
        self.add(block, (rows, cols), trainable)
        off = 0
        for hf_name, r, c in parts:
            assert c == cols
            self.alias[hf_name] = (block, off, (r, c))
            off += r
        del self.alias[block]
        return block

    # ---- allocation
    def allocate(self):
        offs = {}
        for name, s in self.specs.items():
            n = 1
            for d in s['shape']:
                n *= d
            g = s['group']
            o = offs.get(g, 0)
            s['offset'], s['numel'] = o, n
            offs[g] = o + (n + ALIGN - 1) // ALIGN * ALIGN
        self.sizes = offs
        for g, n in offs.items():
            self.flat[g] = torch.zeros(n, dtype=self.dtype, device=self.device)
        for name, s in self.specs.items():
            self.p[name] = self.flat[s['group']][s['offset']:s['offset'] + s['numel']].view(s['shape'])

    def init_training(self):
        """Allocate gradients and fp32 optimizer state for the trainable groups (after weights are loaded)."""
        for g, n in self.sizes.items():
            if g == 'frozen':
                continue
            gd = torch.bfloat16 if (g in ('mat', 'exp') and self.dtype == torch.bfloat16) else torch.float32
            self.gflat[g] = torch.zeros(n, dtype=gd, device=self.device)
            self.master[g] = self.flat[g] if self.dtype == torch.float32 else self.flat[g].to(torch.float32)
            self.m[g] = torch.zeros(n, dtype=torch.float32, device=self.device)
            self.v[g] = torch.zeros(n, dtype=torch.float32, device=self.device)
        for name, s in self.specs.items():
            if s['group'] != 'frozen':
                self.g[name] = self.gflat[s['group']][s['offset']:s['offset'] + s['numel']].view(s['shape'])

    def zero_grad(self):
        # bf16 'mat' gradients are fully overwritten by the dW GEMMs each step; fp32 gradient buffers (emb / vec, and
        # every group in the fp32 parity mode) are accumulated into (Linear.dw: accumulate iff the target is fp32)
        for g, t in self.gflat.items():
            if t.dtype == torch.float32:
                t.zero_()

    def trainable_groups(self):
        return [g for g in ('mat', 'exp', 'emb', 'vec') if g in self.gflat]

    def num_trainable(self) -> int:
        return sum(s['numel'] for s in self.specs.values() if s['group'] != 'frozen')

    def num_params(self) -> int:
        return sum(s['numel'] for s in self.specs.values())

    # ---- HF-format state dict (what save_pretrained / from_pretrained exchange)
    def hf_names(self):
        return list(self.alias)

    def view(self, hf_name) -> torch.Tensor:
        block, off, shape = self.alias[hf_name]
        t = self.p[block]
        return t[off:off + shape[0]] if tuple(t.shape) != tuple(shape) else t

    def grad_view(self, hf_name):
        block, off, shape = self.alias[hf_name]
        if block not in self.g:
            return None
        t = self.g[block]
        return t[off:off + shape[0]] if tuple(t.shape) != tuple(shape) else t

    def load_state_dict(self, sd: dict, strict=True, pad_cols: dict | None = None):
        missing = []
        for hf_name in self.alias:
            if hf_name not in sd:
                missing.append(hf_name)
                continue
            src = sd[hf_name]
            dst = self.view(hf_name)
            if hf_name in self.shard:   # a full checkpoint tensor: keep the rows this rank owns
                first, full = self.shard[hf_name]
                if src.shape[0] == full:
                    src = src[first:first + dst.shape[0]]
            if pad_cols and hf_name in pad_cols:  # zero-padded K (CLIP patch embedding 588 -> 640)
                src = src.reshape(src.shape[0], -1)
                dst.zero_()
                dst[:, :src.shape[1]].copy_(src.to(self.dtype))
            else:
                if tuple(src.shape) != tuple(dst.shape):
                    raise RuntimeError(f'{hf_name}: checkpoint shape {tuple(src.shape)} != {tuple(dst.shape)}')
                dst.copy_(src.to(self.dtype))
        if strict and missing:
            raise RuntimeError(f'missing keys in checkpoint: {missing[:8]}{"..." if len(missing) > 8 else ""}')
        for g in self.master:
            if self.master[g] is not self.flat[g]:
                self.master[g].copy_(self.flat[g])
        return missing

    def opt_state_views(self, hf_name):
        """(master, m, v) fp32 views of one HF-named tensor inside the flat optimizer buffers, or None when it is frozen."""
        block, off, shape = self.alias[hf_name]
        s = self.specs[block]
        if s['group'] not in self.m:
            return None
        out = []
        for flat in (self.master[s['group']], self.m[s['group']], self.v[s['group']]):
            t = flat[s['offset']:s['offset'] + s['numel']].view(s['shape'])
            out.append(t[off:off + shape[0]] if tuple(t.shape) != tuple(shape) else t)
        return tuple(out)

    def load_opt_state(self, m: dict, v: dict, strict=True):
        """Adam moments by HF name (a torch.optim.AdamW / DeepSpeed state re-keyed by parameter name) into the flat fp32 buffers: resuming
        from a foreign optimizer state, and the teacher-forced parity test (tests/test_f32_gpu.py).  After `init_training()`."""
        missing = []
        for hf_name in self.alias:
            views = self.opt_state_views(hf_name)
            if views is None:
                continue
            if hf_name not in m or hf_name not in v:
                missing.append(hf_name)
                continue
            views[1].copy_(m[hf_name].to(torch.float32).reshape(views[1].shape))
            views[2].copy_(v[hf_name].to(torch.float32).reshape(views[2].shape))
        if strict and missing:
            raise RuntimeError(f'missing optimizer state for: {missing[:8]}{"..." if len(missing) > 8 else ""}')
        return missing

    def state_dict(self, unpad: dict | None = None) -> dict:
        out = {}
        for hf_name in self.alias:
            t = self.view(hf_name)
            if unpad and hf_name in unpad:
                shape = unpad[hf_name]
                n = 1
                for d in shape[1:]:
                    n *= d
                t = t[:, :n].reshape(shape)
            out[hf_name] = t.detach().clone()
        return out
