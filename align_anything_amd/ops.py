"""Thin torch-tensor wrappers over the C ABI (include/aa_hip.h).

torch is used here only as the owner of device memory and streams: every function passes raw device
pointers, sizes and the current HIP stream to libaa_hip.so.  Nothing in this module computes on the
CPU or falls back to torch ops.
"""
from __future__ import annotations

import ctypes
import os

import torch

from .lib import call

ACT_NONE, ACT_GELU, ACT_QUICK_GELU, ACT_RELU, ACT_SILU = 0, 1, 2, 3, 4
GEMM_A_T, GEMM_B_N, GEMM_OUT_F32, GEMM_ACCUM = 1, 2, 4, 8
ACT_CODES = {None: 0, 'none': 0, 'gelu': 1, 'quick_gelu': 2, 'relu': 3, 'silu': 4}

bf16 = torch.bfloat16
f32 = torch.float32


def _sfx(t: torch.Tensor, name: str) -> str:
    """'' for the bf16 production kernels, '_f32' for the fp32 parity-mode twins (include/aa_hip_f32.h)."""
    if not t.is_cuda:
        raise RuntimeError(f'{name}: tensor must live on the GPU (the hot path has no CPU fallback)')
    if t.dtype == bf16:
        return ''
    if t.dtype == f32:
        return '_f32'
    raise RuntimeError(f'{name}: expected bfloat16 or float32 activations, got {t.dtype}')


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise RuntimeError(f'{name}: tensor must live on the GPU (the hot path has no CPU fallback)')
    if t.dtype != dtype:
        raise RuntimeError(f'{name}: expected {dtype}, got {t.dtype}')


def _row_major(t: torch.Tensor, name: str):
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError(f'{name}: expected a 2-D tensor with unit inner stride, got {tuple(t.shape)} / {t.stride()}')


# ------------------------------------------------------------------ GEMM
def gemm(a, b, out=None, *, bias=None, residual=None, act=0, a_t=False, b_n=False, out_f32=False,
         accumulate=False):
    """C[M,N] (+)= op(A) @ op(B)^T-ish:  a is [M,K] (or [K,M] when a_t), b is [N,K] (or [K,N] when b_n)."""
    sfx = _sfx(a, 'gemm.a')
    _chk(b, a.dtype, 'gemm.b')
    _row_major(a, 'gemm.a'); _row_major(b, 'gemm.b')
    if a_t:
        K, M = a.shape
    else:
        M, K = a.shape
    if b_n:
        Kb, N = b.shape
    else:
        N, Kb = b.shape
    if K != Kb:
        raise RuntimeError(f'gemm: contraction mismatch {K} vs {Kb}')
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32 if out_f32 else a.dtype, device=a.device)
    else:
        _row_major(out, 'gemm.out')
        if tuple(out.shape) != (M, N):
            raise RuntimeError(f'gemm: out shape {tuple(out.shape)} != {(M, N)}')
    flags = (GEMM_A_T if a_t else 0) | (GEMM_B_N if b_n else 0) | \
            (GEMM_OUT_F32 if out.dtype == torch.float32 else 0) | (GEMM_ACCUM if accumulate else 0)
    ldr = residual.stride(0) if residual is not None else 0
    global _gemm_seq
    prof = GEMM_PROF
    if prof is not None:       # sample every GEMM_PROF_STRIDE-th launch: an event pair around EVERY launch serialises kernel
        _gemm_seq += 1         # boundaries (measured: +2 % step time); a stride coprime to the launches per step stays unbiased
        if _gemm_seq % GEMM_PROF_STRIDE:
            prof = None
    if prof is not None:
        e0 = event_record()
    if sfx and out.dtype != f32:
        raise RuntimeError('gemm: fp32 operands need an fp32 output')
    FLOPS['gemm'] += 2.0 * M * N * K
    # (shapes / layouts aa_gemm_splitk_bf16 does not take stay on the one-launch kernel: K-tile multiple, A^T only with row-major-K B, 8-row multiples under A^T)
    S = 0 if (sfx or K % 64 or (a_t and (not b_n or M % 8)) or ldr % 4) else _splitk_chunks(M, N, K)
    if S:       # few rows against a large weight matrix: the contraction in S chunks side by side (csrc/gemm.hip aa_gemm_splitk_bf16)
        call('aa_gemm_splitk_bf16', a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, a.stride(0), b.stride(0), out.stride(0), _p(bias), _p(residual),
             ldr, int(act), flags, _splitk_ws(S * M * N, a.device).data_ptr(), S, stream())
    else:
        call('aa_gemm_f32' if sfx else 'aa_gemm_bf16', a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, a.stride(0),
             b.stride(0), out.stride(0), _p(bias), _p(residual), ldr, int(act), flags, stream())
    if prof is not None:
        prof.append((e0, event_record(), 2.0 * M * N * K, float(a.element_size()) * (M * K + N * K) + out.element_size() * M * N))
    return out


# Split-K for few-row GEMMs (round 6).  A PPO rollout scores ONE sequence of ~800 tokens with four 7B models and trains on it: every projection is an
# [~800, K] x [N, K]^T launch whose 256 x 128 tiles number 4 x N / 128 -- 112 workgroups for N = 3584, each streaming its slab of the weight matrix alone
# (the down projection, K = 18944: 350 us against a 27 us weight stream; the gate_up input gradient, K = 37888: 800 us).  With the contraction cut into chunks
# that run side by side every compute unit streams weights.  Rule: at most 1024 rows, K >= 1024, fewer than 192 tiles -> enough chunks (<= 8, >= 256 deep) for
# ~450 work items.  AA_GEMM_SPLITK=0: off everywhere (A/B).
SPLITK_ALLOWED = os.environ.get('AA_GEMM_SPLITK', '1') != '0'
SPLITK = False          # on inside `few_row_gemms` scopes only: the PPO / GRPO rollout, scoring forwards and update (their M is one or a few sequences)
_SPLITK_WS = {}


def few_row_gemms(fn):
    """Decorator: ops.gemm may cut few-row launches along K inside `fn` (see _splitk_chunks).  Scoped to the RL trainers' steps rather than switched on for the
    process: the split changes the fp32 association of the accumulator, and the preference trainers' parity envelopes (random-init, chaotic at 32 layers; the MoE
    router's top-k) are pinned with the one-launch kernel."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **kw):
        global SPLITK
        old, SPLITK = SPLITK, SPLITK_ALLOWED
        try:
            return fn(*a, **kw)
        finally:
            SPLITK = old
    return wrapped


def _splitk_chunks(M, N, K):
    if not SPLITK or M > 1024 or K < 1024 or N % 8:
        return 0
    tiles = -(-M // 256) * -(-N // 128)
    if tiles >= 192:
        return 0
    S = min(8, K // 256, -(-448 // tiles))
    return S if S >= 2 else 0


def _splitk_ws(n, device):
    key = str(device)
    ws = _SPLITK_WS.get(key)
    if ws is None or ws.numel() < n:
        ws = _SPLITK_WS[key] = torch.empty(max(n, 1 << 24), dtype=torch.float32, device=device)
    return ws


# executed-work counters (bench.py: MFMA fraction on the FLOPs the step really executes): 2*M*N*K per GEMM launch,
# 4*T*T*hd per (row, head) of attention forward (halved when causal), 2.5x that for its backward
FLOPS = {'gemm': 0.0, 'attn': 0.0}

def gemm_qkv_rope(x, w, pos, cos_t, sin_t, rope_heads, hd, out=None):
    """qkv = x @ w^T with the rotary embedding applied to the first `rope_heads` heads (q and k of the fused [q|k|v] weight): the
    projection GEMM with HF's apply_rotary_pos_emb in its epilogue (aa_gemm_qkv_rope_bf16; unfused pair of kernels when the shape does
    not qualify -- bit-identical either way)."""
    _chk(x, bf16, 'gemm_qkv_rope.x'); _chk(w, bf16, 'gemm_qkv_rope.w')
    _row_major(x, 'gemm_qkv_rope.x'); _row_major(w, 'gemm_qkv_rope.w')
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=bf16, device=x.device) if out is None else out
    FLOPS['gemm'] += 2.0 * M * N * K
    prof = _prof_begin()
    call('aa_gemm_qkv_rope_bf16', x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, x.stride(0), w.stride(0), out.stride(0),
         pos.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), int(rope_heads) * hd, hd, stream())
    _prof_end(prof, 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N))
    return out


def gemm_glu_fwd(x, w_gu, F):
    """LlamaMLP front half in one launch: returns (gate_up [M, 2F], act [M, F] = silu(gate) * up)."""
    _chk(x, bf16, 'gemm_glu_fwd.x'); _chk(w_gu, bf16, 'gemm_glu_fwd.w')
    _row_major(x, 'gemm_glu_fwd.x'); _row_major(w_gu, 'gemm_glu_fwd.w')
    M, K = x.shape
    gu = torch.empty((M, 2 * F), dtype=bf16, device=x.device)
    act = torch.empty((M, F), dtype=bf16, device=x.device)
    FLOPS['gemm'] += 4.0 * M * F * K
    prof = _prof_begin()
    call('aa_gemm_glu_fwd_bf16', x.data_ptr(), w_gu.data_ptr(), gu.data_ptr(), act.data_ptr(), M, F, K, x.stride(0), w_gu.stride(0),
         gu.stride(0), act.stride(0), stream())
    _prof_end(prof, 4.0 * M * F * K, 2.0 * (M * K + 2 * F * K + 3 * M * F))
    return gu, act


# d_act workspace of the unfused SwiGLU-backward pair: ONE grow-only flat buffer per device, sized to the largest M x F seen (variable sequence
# lengths give M many values; a buffer per shape would grow without bound -- ADVICE r3).  A buffer that only ever served probes whose winner was the
# fused kernel is dropped again.
_GLU_WS = {}          # device -> flat bf16 buffer
_GLU_WS_KEEP = set()  # devices on which the unfused plan has actually run
GLU_BWD_PROBE = os.environ.get('AA_GLU_BWD_PROBE', '1') != '0'
GLU_BWD_PROBE_LOG = []   # (M, F, K, ms_fused, ms_unfused) of every probe this process ran (bench.py reports it)


def _glu_ws(M, F, device, keep=True):
    key = str(device)
    ws = _GLU_WS.get(key)
    if ws is None or ws.numel() < M * F:
        ws = _GLU_WS[key] = torch.empty(M * F, dtype=bf16, device=device)
    if keep:
        _GLU_WS_KEEP.add(key)
    return ws[:M * F].view(M, F)


def gemm_glu_bwd(dy, w_down, gu, F):
    """d[gate|up] [M, 2F] of the block above from dy [M, h] (gradient of the down projection's output) and w_down [h, F].  The C side
    owns the decision between the fused epilogue and the unfused GEMM + aa_swiglu_bwd pair (aa_gemm_glu_bwd_plan); the first time a
    fusable shape is seen both are timed on the live buffers (aa_gemm_glu_bwd_probe: same bits either way) and the faster one is kept
    for the process -- the fused epilogue is 1.6 x slower than the pair on boxes with a long memory round trip."""
    _chk(dy, bf16, 'gemm_glu_bwd.dy'); _chk(w_down, bf16, 'gemm_glu_bwd.w'); _chk(gu, bf16, 'gemm_glu_bwd.gu')
    _row_major(dy, 'gemm_glu_bwd.dy'); _row_major(w_down, 'gemm_glu_bwd.w'); _row_major(gu, 'gemm_glu_bwd.gu')
    M, K = dy.shape
    dgu = torch.empty_like(gu)
    plan = ctypes.c_int(-1)
    args = (dy.data_ptr(), w_down.data_ptr(), gu.data_ptr(), dgu.data_ptr())
    dims = (M, F, K, dy.stride(0), w_down.stride(0), gu.stride(0), dgu.stride(0))
    call('aa_gemm_glu_bwd_plan', *args, *dims, ctypes.addressof(plan))
    FLOPS['gemm'] += 2.0 * M * F * K
    if plan.value == 2 and GLU_BWD_PROBE:
        t = (ctypes.c_float * 2)()
        call('aa_gemm_glu_bwd_probe', *args, _glu_ws(M, F, dy.device, keep=False).data_ptr(), *dims, 3, ctypes.addressof(t), ctypes.addressof(t) + 4, stream())
        GLU_BWD_PROBE_LOG.append((M, F, K, float(t[0]), float(t[1])))
        if float(t[0]) <= float(t[1]) and str(dy.device) not in _GLU_WS_KEEP:
            _GLU_WS.pop(str(dy.device), None)       # the fused kernel won (the probe synchronised the stream): nothing needs the workspace
        return dgu
    ws = _glu_ws(M, F, dy.device) if plan.value == 0 else None
    prof = _prof_begin()
    call('aa_gemm_glu_bwd_bf16', *args, _p(ws), *dims, stream())
    _prof_end(prof, 2.0 * M * F * K, 2.0 * (M * K + F * K + 4 * M * F))
    return dgu


_FUSE = os.environ.get('AA_GEMM_FUSE', '1') != '0'


def fuse_enabled() -> bool:
    return _FUSE


_ATTN_ROPE = os.environ.get('AA_ATTN_ROPE', '1') != '0'


def attn_rope_fused() -> bool:
    """bf16 backward: the rotary backward of dQ / dK inside the attention backward kernels (aa_attn_bwd_rope) instead of its own launch;
    AA_ATTN_ROPE=0 keeps the two launches (same bits -- tests/test_attention_gpu.py -- the switch exists for same-box timing)."""
    return _FUSE and _ATTN_ROPE


def gemm_set_fuse(on: bool) -> None:
    global _FUSE
    _FUSE = bool(on)
    call('aa_gemm_set_fuse', int(bool(on)))


def _prof_begin():
    global _gemm_seq
    prof = GEMM_PROF
    if prof is not None:
        _gemm_seq += 1
        if _gemm_seq % GEMM_PROF_STRIDE:
            prof = None
    return (prof, event_record()) if prof is not None else None


def _prof_end(tok, flops, nbytes):
    if tok is not None:
        tok[0].append((tok[1], event_record(), flops, nbytes))


# HIP events on the launch stream (bench.py roofline: per-launch GEMM durations over the timed region)
GEMM_PROF = None
GEMM_PROF_STRIDE = 1
_gemm_seq = 0


# The same for the kernels below the GEMMs ("fraction of HBM peak per memory-bound kernel"): bench.py sets KPROF = {} and
# every KPROF_STRIDE-th launch of a kernel kind (every launch of the kinds in KPROF_EVERY: three per step) carries an event pair on its launch stream, with
# the ALGORITHMIC bytes (and FLOPs for attention) of that launch.  None = off: one `is None` test per call.
KPROF = None
KPROF_STRIDE = 7
KPROF_EVERY = ('adamw_flat', 'grad_sumsq')
_kseq: dict = {}


def _kprof_begin(kind):
    if KPROF is None:
        return None
    n = _kseq[kind] = _kseq.get(kind, 0) + 1
    if kind not in KPROF_EVERY and n % KPROF_STRIDE:
        return None
    return (kind, event_record())


def _kprof_end(tok, nbytes, flops=0.0):
    if tok is not None:
        KPROF.setdefault(tok[0], []).append((tok[1], event_record(), float(nbytes), float(flops)))


EVENT_POOL: list = []   # pre-created HIP events (bench.py fills it so that creation stays out of the timed region)


def event_pool_fill(n: int) -> None:
    import ctypes
    for _ in range(max(0, n - len(EVENT_POOL))):
        ev = ctypes.c_void_p()
        call('aa_event_create', ctypes.byref(ev))
        EVENT_POOL.append(ev)


def event_record():
    import ctypes
    if EVENT_POOL:
        ev = EVENT_POOL.pop()
    else:
        ev = ctypes.c_void_p()
        call('aa_event_create', ctypes.byref(ev))
    call('aa_event_record', ev, stream())
    return ev


def event_elapsed_ms(e0, e1) -> float:
    import ctypes
    ms = ctypes.c_float()
    call('aa_event_elapsed_ms', e0, e1, ctypes.byref(ms))
    return float(ms.value)


def gemm_set_tile(tile: int) -> None:
    call('aa_gemm_set_tile', int(tile))


# ------------------------------------------------------------------ norms
NORM_WS_ROWS = 512
_ws_cache: dict = {}


def _norm_ws(device, h, planes):
    """Scratch for the per-workgroup dw/db partial rows of the norm backward kernels (stream-ordered reuse)."""
    key = (device, h, planes)
    t = _ws_cache.get(key)
    if t is None:
        t = torch.empty((planes * NORM_WS_ROWS, h), dtype=torch.float32, device=device)
        _ws_cache[key] = t
    return t


def rmsnorm_fwd(x, w, eps, out=None, rstd=None):
    rows, h = x.shape
    out = torch.empty_like(x) if out is None else out
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if rstd is None else rstd
    tok = _kprof_begin('rmsnorm_fwd')
    call('aa_rmsnorm_fwd' + _sfx(x, 'rmsnorm_fwd'), x.data_ptr(), w.data_ptr(), out.data_ptr(), rstd.data_ptr(), rows, h, float(eps), stream())
    _kprof_end(tok, x.element_size() * (2 * rows * h + h) + 4 * rows)            # x read, y written, the weight row, rstd
    return out, rstd


def rmsnorm_rope_fwd(x, w, eps, pos, cos_t, sin_t, heads, hd=None):
    """rope(rmsnorm(x)) per head in one pass (Qwen3 q_norm / k_norm + rotary embedding); bit-identical to rmsnorm_fwd + rope_.
    x: [tokens * heads, hd] dense, or -- with hd given -- [tokens, heads * hd] as a COLUMN SLICE of a wider row-major buffer (the fused q | k | v
    projection output; x.stride(0) = its width).  Returns (y [tokens * heads, hd] dense, rstd [tokens * heads])."""
    if hd is None:
        rows, hd = x.shape
        ldx = heads * hd
    else:
        if x.stride(1) != 1 or x.shape[1] != heads * hd:
            raise RuntimeError(f'rmsnorm_rope_fwd: expected a [tokens, {heads * hd}] column slice, got {tuple(x.shape)} strides {x.stride()}')
        rows, ldx = x.shape[0] * heads, x.stride(0)
    if cos_t.dtype != x.dtype:
        raise RuntimeError(f'rmsnorm_rope_fwd: tables are {cos_t.dtype}, activations {x.dtype}')
    out = torch.empty((rows, hd), dtype=x.dtype, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    call('aa_rmsnorm_rope_fwd' + _sfx(x, 'rmsnorm_rope_fwd'), x.data_ptr(), int(ldx), w.data_ptr(), out.data_ptr(), rstd.data_ptr(), rows, hd, float(eps), pos.data_ptr(),
         cos_t.data_ptr(), sin_t.data_ptr(), int(heads), stream())
    return out, rstd


def rmsnorm_heads_bwd(dy, x, w, rstd, dw, dx, heads, hd):
    """Backward of the per-head norm with x [tokens, heads * hd] (the saved projection output) and dx as column slices of the fused q | k | v buffers;
    dy [tokens * heads, hd] dense.  dx is written (not accumulated)."""
    rows = x.shape[0] * heads
    if x.stride(1) != 1 or dx.stride(1) != 1 or tuple(dx.shape) != tuple(x.shape):
        raise RuntimeError('rmsnorm_heads_bwd: x / dx must be [tokens, heads * hd] column slices')
    ws = _norm_ws(x.device, hd, 1) if dw is not None else None
    call('aa_rmsnorm_heads_bwd' + _sfx(x, 'rmsnorm_heads_bwd'), dy.data_ptr(), x.data_ptr(), x.stride(0), w.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dx.stride(0),
         _p(dw), _p(ws), NORM_WS_ROWS, rows, hd, int(heads), stream())
    return dx


def rmsnorm_bwd(dy, x, w, rstd, dw, dx=None, add_to_dx=False):
    rows, h = x.shape
    dx = torch.empty_like(x) if dx is None else dx
    ws = _norm_ws(x.device, h, 1) if dw is not None else None
    tok = _kprof_begin('rmsnorm_bwd')
    call('aa_rmsnorm_bwd' + _sfx(x, 'rmsnorm_bwd'), dy.data_ptr(), x.data_ptr(), w.data_ptr(), rstd.data_ptr(), dx.data_ptr(), _p(dw),
         _p(ws), NORM_WS_ROWS, rows, h, int(add_to_dx), stream())
    _kprof_end(tok, x.element_size() * ((4 if add_to_dx else 3) * rows * h + h) + 4 * rows)      # x, dy (and the residual gradient) read, dx written
    return dx


def layernorm_fwd(x, w, b, eps, out=None, want_stats=True):
    rows, h = x.shape
    out = torch.empty_like(x) if out is None else out
    mean = rstd = None
    if want_stats:
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    call('aa_layernorm_fwd' + _sfx(x, 'layernorm_fwd'), x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), _p(mean), _p(rstd),
         rows, h, float(eps), stream())
    return out, mean, rstd


def layernorm_bwd(dy, x, w, mean, rstd, dw, db, dx=None, add_to_dx=False):
    rows, h = x.shape
    dx = torch.empty_like(x) if dx is None else dx
    ws = _norm_ws(x.device, h, 2) if (dw is not None or db is not None) else None
    call('aa_layernorm_bwd' + _sfx(x, 'layernorm_bwd'), dy.data_ptr(), x.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
         dx.data_ptr(), _p(dw), _p(db), _p(ws), NORM_WS_ROWS, rows, h, int(add_to_dx), stream())
    return dx


def rowdot_fwd(x, w):
    rows, h = x.shape
    out = torch.empty(rows, dtype=torch.float32, device=x.device)
    call('aa_rowdot_fwd' + _sfx(x, 'rowdot_fwd'), x.data_ptr(), w.data_ptr(), out.data_ptr(), rows, h, stream())
    return out


def rowdot_bwd(dy, x, w, dw):
    rows, h = x.shape
    dx = torch.empty_like(x)
    ws = _norm_ws(x.device, h, 1) if dw is not None else None
    call('aa_rowdot_bwd' + _sfx(x, 'rowdot_bwd'), dy.data_ptr(), x.data_ptr(), w.data_ptr(), dx.data_ptr(), _p(dw), _p(ws), NORM_WS_ROWS,
         rows, h, stream())
    return dx


# ------------------------------------------------------------------ pointwise
def rope_(buf, col0, nheads, hd, pos, cos_t, sin_t, inverse=False, head_stride=0, precise=False):
    rows = buf.shape[0]
    if precise and (cos_t.dtype != f32 or sin_t.dtype != f32):
        raise RuntimeError('rope_: precise mode takes fp32 cos/sin tables')
    if not precise and cos_t.dtype != buf.dtype:
        raise RuntimeError(f'rope_: tables are {cos_t.dtype}, activations {buf.dtype}')
    call('aa_rope_inplace' + _sfx(buf, 'rope_'), buf.data_ptr(), buf.stride(0), int(col0), int(nheads), int(hd), pos.data_ptr(),
         cos_t.data_ptr(), sin_t.data_ptr(), rows, int(inverse), int(head_stride), int(precise), stream())
    return buf


def mrope_tables(pos3, inv_freq, sections, dtype):
    """pos3 int32 [3, rows] -> (cos, sin) [rows, half] for multimodal RoPE (sections = mrope_section, sums to half)."""
    rows = pos3.shape[1]
    half = inv_freq.numel()
    cos = torch.empty((rows, half), dtype=dtype, device=pos3.device)
    sin = torch.empty_like(cos)
    call('aa_mrope_tables' + _sfx(cos, 'mrope_tables'), pos3.data_ptr(), rows, inv_freq.data_ptr(), half, int(sections[0]),
         int(sections[1]), cos.data_ptr(), sin.data_ptr(), stream())
    return cos, sin


def swiglu_fwd(gate_up, out=None):
    M, F2 = gate_up.shape
    F = F2 // 2
    out = torch.empty((M, F), dtype=gate_up.dtype, device=gate_up.device) if out is None else out
    call('aa_swiglu_fwd' + _sfx(gate_up, 'swiglu_fwd'), gate_up.data_ptr(), out.data_ptr(), M, F, stream())
    return out


def swiglu_bwd(gate_up, dact, out=None):
    M, F2 = gate_up.shape
    out = torch.empty_like(gate_up) if out is None else out
    tok = _kprof_begin('swiglu_bwd')
    call('aa_swiglu_bwd' + _sfx(gate_up, 'swiglu_bwd'), gate_up.data_ptr(), dact.data_ptr(), out.data_ptr(), M, F2 // 2, stream())
    _kprof_end(tok, gate_up.element_size() * M * (F2 // 2) * 5)                  # gate | up and d act read, d gate | d up written
    return out


def act_fwd(x, act, out=None):
    out = torch.empty_like(x) if out is None else out
    call('aa_act_fwd' + _sfx(x, 'act_fwd'), x.data_ptr(), out.data_ptr(), x.numel(), int(act), stream())
    return out


def act_bwd(pre, dy, act, out=None):
    out = torch.empty_like(pre) if out is None else out
    call('aa_act_bwd' + _sfx(pre, 'act_bwd'), pre.data_ptr(), dy.data_ptr(), out.data_ptr(), pre.numel(), int(act), stream())
    return out


def add(a, b, out=None):
    out = torch.empty_like(a) if out is None else out
    call('aa_add' + _sfx(a, 'add'), a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), stream())
    return out


# ------------------------------------------------------------------ embedding
def image_slot_index(ids_flat, image_token_id):
    n = ids_flat.numel()
    slot = torch.empty(n, dtype=torch.int32, device=ids_flat.device)
    count = torch.empty(1, dtype=torch.int32, device=ids_flat.device)
    call('aa_image_slot_index', ids_flat.data_ptr(), n, int(image_token_id), slot.data_ptr(), count.data_ptr(), stream())
    return slot, count


def embed_fwd(ids_flat, E, slot=None, feat=None, pos=None, P=None):
    n = ids_flat.numel()
    vocab, h = E.shape
    out = torch.empty((n, h), dtype=E.dtype, device=E.device)
    call('aa_embed_fwd' + _sfx(E, 'embed_fwd'), ids_flat.data_ptr(), _p(slot), E.data_ptr(), _p(feat), _p(pos), _p(P), out.data_ptr(),
         n, h, vocab, stream())
    return out


def embed_bwd(ids_flat, dx, vocab, slot=None, pos=None, dE=None, dfeat=None, dP=None):
    n, h = dx.shape
    call('aa_embed_bwd' + _sfx(dx, 'embed_bwd'), ids_flat.data_ptr(), _p(slot), _p(pos), dx.data_ptr(), _p(dE), _p(dfeat), _p(dP), n, h,
         int(vocab), stream())


def transpose(x, out=None, pad_cols_to=None):
    """out[C, R(+pad)] = x[R, C]^T ; padding columns (if any) must already be zero in `out`."""
    R, C = x.shape
    if out is None:
        Rp = R if pad_cols_to is None else pad_cols_to
        out = torch.zeros((C, Rp), dtype=x.dtype, device=x.device) if Rp != R else \
            torch.empty((C, R), dtype=x.dtype, device=x.device)
    call('aa_transpose_f32' if _sfx(x, 'transpose') else 'aa_transpose_bf16', x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), R, C, stream())
    return out


def colsum_(x, out_f32):
    R, C = x.shape
    call('aa_colsum_f32' if _sfx(x, 'colsum_') else 'aa_colsum_bf16', x.data_ptr(), x.stride(0), R, C, out_f32.data_ptr(), stream())
    return out_f32


def patch_im2col(pixels, patch, Kp, dtype=bf16):
    n_img, ch, H, W = pixels.shape
    G = H // patch
    out = torch.empty((n_img * G * G, Kp), dtype=dtype, device=pixels.device)
    dt = 0 if pixels.dtype == bf16 else 1
    if pixels.dtype not in (bf16, torch.float32):
        raise RuntimeError(f'patch_im2col: pixel dtype {pixels.dtype} not supported')
    call('aa_patch_im2col' + _sfx(out, 'patch_im2col'), pixels.data_ptr(), dt, out.data_ptr(), n_img, ch, H, int(patch), int(Kp), stream())
    return out


def _rows64(n):
    return (n + 63) // 64 * 64


def conv1d_im2col(x, B, C, Tin, stride, channels_first, dtype):
    """Conv1d(k=3, padding=1) patches: x = [B, C, Tin] features (channels_first) or token-major [>=B*Tin, C] -> [B*Tout, 3C]
    (rows padded to a multiple of 64 with zeros: they are the contraction dim of the weight-gradient GEMM)."""
    Tout = (Tin + 2 - 3) // stride + 1
    col = torch.zeros((_rows64(B * Tout), 3 * C), dtype=dtype, device=x.device)
    sb, sc, st_ = (C * Tin, Tin, 1) if channels_first else (Tin * C, 1, C)
    if x.dtype not in (bf16, f32) or not x.is_contiguous():
        raise RuntimeError(f'conv1d_im2col: contiguous bf16 / fp32 input expected, got {x.dtype}')
    call('aa_conv1d_im2col' + _sfx(col, 'conv1d_im2col'), x.data_ptr(), 0 if x.dtype == bf16 else 1, sb, sc, st_, col.data_ptr(), B, C, Tin,
         Tout, int(stride), stream())
    return col, Tout


def conv1d_col2im(dcol, B, C, Tin, Tout, stride):
    dx = torch.zeros((_rows64(B * Tin), C), dtype=dcol.dtype, device=dcol.device)
    call('aa_conv1d_col2im' + _sfx(dcol, 'conv1d_col2im'), dcol.data_ptr(), dx.data_ptr(), B, C, Tin, Tout, int(stride), stream())
    return dx


def avgpool2(x, rows_out, backward=False):
    """forward: x [>= 2*rows_out, C] -> [rows64(rows_out), C]; backward: x = dy [>= rows_out, C] -> dx [rows64(2*rows_out), C]."""
    C = x.shape[1]
    y = torch.zeros((_rows64(2 * rows_out if backward else rows_out), C), dtype=x.dtype, device=x.device)
    call('aa_avgpool2' + _sfx(x, 'avgpool2'), x.data_ptr(), y.data_ptr(), rows_out, C, int(backward), stream())
    return y


def clip_embed(patch_emb, cls, pos, n_img, G2):
    h = patch_emb.shape[1]
    out = torch.empty((n_img * (G2 + 1), h), dtype=patch_emb.dtype, device=patch_emb.device)
    call('aa_clip_embed' + _sfx(patch_emb, 'clip_embed'), patch_emb.data_ptr(), cls.data_ptr(), pos.data_ptr(), out.data_ptr(), n_img, G2, h, stream())
    return out


def f32_to_bf16(x, out=None):
    out = torch.empty(x.shape, dtype=bf16, device=x.device) if out is None else out
    call('aa_f32_to_bf16', x.data_ptr(), out.data_ptr(), x.numel(), stream())
    return out


# ------------------------------------------------------------------ mixture of experts
def moe_route(logits, k, norm_topk):
    rows, E = logits.shape
    dev = logits.device
    probs = torch.empty((rows, E), dtype=f32, device=dev)
    idx = torch.empty((rows, k), dtype=torch.int32, device=dev)
    w = torch.empty((rows, k), dtype=logits.dtype, device=dev)
    call('aa_moe_route' + _sfx(logits, 'moe_route'), logits.data_ptr(), logits.stride(0), rows, E, int(k), int(bool(norm_topk)),
         probs.data_ptr(), idx.data_ptr(), w.data_ptr(), stream())
    return probs, idx, w


def moe_route_bwd(probs, idx, dw, norm_topk, dtype):
    rows, E = probs.shape
    dlogits = torch.empty((rows, E), dtype=dtype, device=probs.device)
    call('aa_moe_route_bwd' + _sfx(dlogits, 'moe_route_bwd'), probs.data_ptr(), idx.data_ptr(), dw.data_ptr(), rows, E, idx.shape[1],
         int(bool(norm_topk)), dlogits.data_ptr(), dlogits.stride(0), stream())
    return dlogits


# alignment of the expert segments = the largest row tile of the grouped GEMMs: 256 (the row-grouped launches run on gemm4's 256-row tile, csrc/gemm4.hip GRP;
# 25 % pad rows at 512 rows per expert against 12.5 % at 128, and still 6.5 % faster per step); AA_MOE_GEMM4=0: 128, the 8-wave kernel's tile (same-box A/B)
MOE_ALIGN = 128 if os.environ.get('AA_MOE_GEMM4', '1') == '0' else 256
MOE_TILE_ROWS = 128     # rows per entry of the plan's tile table, whatever the alignment


def moe_plan(idx, E, align=MOE_ALIGN, allow_invalid=False):
    """Device-side expert-major layout (no host read).  Returns dict(pos [rows,k], src [cap], tile_expert, off [E+1], counts [E], cap).
    align = MOE_ALIGN: segments padded to the grouped GEMM's row tile; align = 1: the dense expert-major order (the send buffer
    of the expert-parallel exchange).  allow_invalid: entries of idx outside [0, E) (-1 = a row of the capacity-padded exchange that
    carries no token) belong to no expert: pos = -1, which moe_combine / moe_combine_bwd skip."""
    rows, k = idx.shape
    dev = idx.device
    cap = (rows * k + E * (align - 1) + align - 1) // align * align
    i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)
    pos = torch.full((rows * k,), -1, dtype=torch.int32, device=dev) if allow_invalid else i32(rows * k)
    plan = {'pos': pos.view(rows, k), 'src': i32(cap), 'tile_expert': i32(cap // (MOE_TILE_ROWS if align % MOE_TILE_ROWS == 0 else align)), 'off': i32(E + 1), 'counts': i32(E),
            'cap': cap, 'E': E, 'align': align}
    call('aa_moe_plan', idx.data_ptr(), rows, k, E, align, cap, plan['counts'].data_ptr(), plan['off'].data_ptr(), plan['pos'].data_ptr(),
         plan['src'].data_ptr(), plan['tile_expert'].data_ptr(), stream())
    return plan


def grouped_out(rows, n, like):
    """Output buffer of a grouped GEMM over the padded expert-major layout: the bf16 kernel writes every tile (zeros where a tile has no expert), so no
    memset; the fp32 twin kernel skips those tiles and gets a zeroed buffer."""
    return (torch.empty if like.dtype == bf16 else torch.zeros)((rows, n), dtype=like.dtype, device=like.device)


def gemm_grouped(a, w3, plan, out=None, b_n=False):
    """Rows grouped by expert: out[cap, N] = a[cap, K] @ op(w3[e]) for every 128-row tile's expert e (w3 = [E, N, K], or [E, K, N] with b_n)."""
    sfx = _sfx(a, 'gemm_grouped')
    cap, K = a.shape
    E = w3.shape[0]
    N = w3.shape[2] if b_n else w3.shape[1]
    if (w3.shape[1] if b_n else w3.shape[2]) != K or w3.dtype != a.dtype or not w3.is_contiguous():
        raise RuntimeError(f'gemm_grouped: weight {tuple(w3.shape)} does not match activations {tuple(a.shape)}')
    out = grouped_out(cap, N, a) if out is None else out
    call('aa_gemm_grouped' + (sfx or '_bf16'), a.data_ptr(), w3.data_ptr(), out.data_ptr(), cap, N, K, a.stride(0), w3.stride(1), out.stride(0),
         GEMM_B_N if b_n else 0, 3 if plan.get('align', 1) % 256 == 0 else 1, plan['tile_expert'].data_ptr(), None, w3.stride(0), E, stream())
    return out


def gemm_grouped_dw(dy, x, plan, out3, accumulate=False):
    """Per-expert weight gradients in one launch: out3[e] (+)= dy[seg_e]^T @ x[seg_e]   (out3 = [E, N_out, K_in])."""
    sfx = _sfx(dy, 'gemm_grouped_dw')
    E, M, N = out3.shape
    if dy.shape[1] != M or x.shape[1] != N or not out3.is_contiguous():
        raise RuntimeError('gemm_grouped_dw: shape mismatch')
    if sfx and out3.dtype != f32:
        raise RuntimeError('gemm_grouped_dw: fp32 operands need fp32 gradients')
    flags = GEMM_A_T | GEMM_B_N | (GEMM_OUT_F32 if out3.dtype == f32 and not sfx else 0) | (GEMM_ACCUM if accumulate else 0)
    call('aa_gemm_grouped' + (sfx or '_bf16'), dy.data_ptr(), x.data_ptr(), out3.data_ptr(), M, N, 0, dy.stride(0), x.stride(0), out3.stride(1),
         flags, 2, None, plan['off'].data_ptr(), out3.stride(0), E, stream())
    return out3


def moe_gather(x, src_row):
    rows_out, h = src_row.numel(), x.shape[1]
    out = torch.empty((rows_out, h), dtype=x.dtype, device=x.device)
    if rows_out == 0:
        return out
    call('aa_moe_gather' + _sfx(x, 'moe_gather'), x.data_ptr(), src_row.data_ptr(), out.data_ptr(), rows_out, h, stream())
    return out


def gather2_add(x, row_a, row_b):
    """out[r] = x[row_a[r]] + x[row_b[r]] (-1: zero row), one rounding: == add(moe_gather(x, row_a), moe_gather(x, row_b)) bit for bit, in one pass."""
    rows_out, h = row_a.numel(), x.shape[1]
    out = torch.empty((rows_out, h), dtype=x.dtype, device=x.device)
    if rows_out == 0:
        return out
    call('aa_gather2_add' + _sfx(x, 'gather2_add'), x.data_ptr(), row_a.data_ptr(), row_b.data_ptr(), out.data_ptr(), rows_out, h, stream())
    return out


def moe_combine(yp, pos, w, rows, residual=None):
    k, h = pos.shape[1], yp.shape[1]
    out = torch.empty((rows, h), dtype=yp.dtype, device=yp.device)
    if rows == 0:
        return out
    call('aa_moe_combine' + _sfx(yp, 'moe_combine'), yp.data_ptr(), pos.data_ptr(), _p(w), _p(residual), out.data_ptr(), rows, k, h, stream())
    return out


def moe_combine_bwd(dout, yp, pos, w, src=None):
    """src = the plan's row -> token table (-1 = pad row): the kernel zeroes the pad rows of dyp itself; without it the whole buffer is memset first."""
    rows, k = pos.shape
    dyp = torch.empty_like(yp) if src is not None else torch.zeros_like(yp)      # pad rows of every expert segment carry no gradient
    dw = torch.empty((rows, k), dtype=f32, device=yp.device)
    if src is not None and src.numel() != yp.shape[0]:
        raise RuntimeError(f'moe_combine_bwd: src has {src.numel()} rows, the layout {yp.shape[0]}')
    call('aa_moe_combine_bwd' + _sfx(yp, 'moe_combine_bwd'), dout.data_ptr(), yp.data_ptr(), pos.data_ptr(), w.data_ptr(), dyp.data_ptr(),
         dw.data_ptr(), rows, k, yp.shape[1], _p(src), yp.shape[0], stream())
    return dyp, dw


# ------------------------------------------------------------------ attention
def attn_fwd(q, k, v, N, T, H, Hkv, hd, causal, scale, start=None, out=None, kv_len=None, q_skip=None, work_frac=1.0):
    """q/k/v: 2-D views [N*T, >=H*hd] (column slices of the fused qkv buffer are fine).
    q_skip (int32 [N], bf16 only): query rows below it have no consumer (aa_attn_fwd_qskip) -- their O / lse rows may stay unwritten."""
    out = torch.empty((N * T, H * hd), dtype=q.dtype, device=q.device) if out is None else out
    lse = torch.empty((N, H, T), dtype=torch.float32, device=q.device)
    fl = 4.0 * N * H * T * T * hd * (0.5 if causal else 1.0) * work_frac      # work_frac: the share of the causal triangle a q_skip leaves (host arithmetic of the plan)
    FLOPS['attn'] += fl
    tok = _kprof_begin('attn_fwd' if (hd == 128 and causal) else 'attn_fwd_other')
    if q_skip is not None and q.dtype == bf16:
        call('aa_attn_fwd_qskip', q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), _p(start), _p(kv_len),
             q.stride(0), k.stride(0), v.stride(0), out.stride(0), N, T, H, Hkv, hd, int(causal), float(scale), q_skip.data_ptr(), stream())
    else:
        call('aa_attn_fwd' + _sfx(q, 'attn_fwd'), q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), _p(start), _p(kv_len),
             q.stride(0), k.stride(0), v.stride(0), out.stride(0), N, T, H, Hkv, hd, int(causal), float(scale), stream())
    _kprof_end(tok, q.element_size() * N * T * hd * (2 * H + 2 * Hkv) + 4 * N * H * T, fl)
    return out, lse


def attn_bwd(q, k, v, o, do, lse, dq, dk, dv, N, T, H, Hkv, hd, causal, scale, start=None, kv_len=None, rope=None, q_skip=None, work_frac=1.0):
    """rope = (pos int32 [rows], cos_t, sin_t bf16 [., hd / 2]): dq and dk leave the kernels already rotated back (the backward of the rotary embedding
    of the forward, `rope_(..., inverse=True)`), bf16 only -- bit-identical to the separate launch."""
    delta = torch.empty((N, H, T), dtype=torch.float32, device=q.device)
    fl = 10.0 * N * H * T * T * hd * (0.5 if causal else 1.0) * work_frac       # the five matmuls of the algorithm (S, dP, dV, dK, dQ)
    FLOPS['attn'] += fl
    tok = _kprof_begin('attn_bwd' if (hd == 128 and causal) else 'attn_bwd_other')
    args = (q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
            delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), _p(start), _p(kv_len), q.stride(0), k.stride(0),
            v.stride(0), o.stride(0), do.stride(0), dq.stride(0), dk.stride(0), dv.stride(0), N, T, H, Hkv, hd, int(causal), float(scale))
    if rope is not None:
        pos, cos_t, sin_t = rope
        if q.dtype != bf16 or cos_t.dtype != bf16 or pos.dtype != torch.int32 or pos.numel() < N * T:
            raise RuntimeError('attn_bwd(rope=): bf16 activations and tables, int32 positions for every token row')
    if q_skip is not None and q.dtype == bf16:       # (the caller's do is zero on the skipped query rows)
        r3 = (None, None, None) if rope is None else (rope[0].data_ptr(), rope[1].data_ptr(), rope[2].data_ptr())
        call('aa_attn_bwd_qskip', *args, *r3, q_skip.data_ptr(), stream())
    elif rope is not None:
        call('aa_attn_bwd_rope', *args, pos.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), stream())
    else:
        call('aa_attn_bwd' + _sfx(q, 'attn_bwd'), *args, stream())
    _kprof_end(tok, q.element_size() * N * T * hd * (4 * H + 4 * Hkv) + 8 * N * H * T, fl)      # q, o, do read, dq written; k, v read, dk, dv written; lse, delta
    return dq, dk, dv


# ------------------------------------------------------------------ RLHF math
def logprob_gather_fwd(logits, labels, round_bf16=False):
    rows, V = logits.shape
    dt = 0 if logits.dtype == bf16 else 1
    logp = torch.empty(rows, dtype=torch.float32, device=logits.device)
    lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
    call('aa_logprob_gather_fwd', logits.data_ptr(), logits.stride(0), labels.data_ptr(), logp.data_ptr(),
         lse.data_ptr(), rows, V, dt, int(round_bf16), stream())
    return logp, lse


def logprob_gather_bwd(logits, labels, lse, dlogp, out=None):
    rows, V = logits.shape
    dt = 0 if logits.dtype == bf16 else 1
    out = torch.empty_like(logits) if out is None else out
    call('aa_logprob_gather_bwd', logits.data_ptr(), logits.stride(0), labels.data_ptr(), lse.data_ptr(),
         dlogp.data_ptr(), out.data_ptr(), out.stride(0), rows, V, dt, stream())
    return out


LMHEAD_CHUNK = int(os.environ.get('AA_LMHEAD_CHUNK', '8192'))     # vocabulary columns per pass of the fused lm_head x log-prob


def lmhead_fused() -> bool:
    """AA_LMHEAD_FUSED=0 keeps the unfused pair (lm_head GEMM into a [rows, V] buffer, then aa_logprob_gather_*) for A/B and parity runs."""
    return os.environ.get('AA_LMHEAD_FUSED', '1') != '0'


def _lmhead_ws_bytes(rows, chunk, h, dt, backward) -> int:
    import ctypes
    nb = ctypes.c_long(0)
    call('aa_lmhead_logprob_ws_bytes', rows, chunk, h, dt, backward, ctypes.byref(nb))
    return int(nb.value)


def lmhead_logprob_fwd(hidden, w, labels, round_bf16=False, chunk=None):
    """log_softmax(hidden @ w^T).gather(labels) per row without the [rows, V] logits buffer (aa_lmhead_logprob_fwd): the vocabulary is
    walked `chunk` columns at a time through a [rows, chunk] scratch.  Bit-identical to gemm + logprob_gather_fwd."""
    _chk(w, hidden.dtype, 'lmhead_logprob_fwd.w'); _row_major(hidden, 'lmhead_logprob_fwd.hidden'); _row_major(w, 'lmhead_logprob_fwd.w')
    rows, h = hidden.shape
    V = w.shape[0]
    chunk = int(chunk or LMHEAD_CHUNK)
    dt = 0 if hidden.dtype == bf16 else 1
    nb = _lmhead_ws_bytes(rows, chunk, h, dt, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device=hidden.device)
    logp = torch.empty(rows, dtype=torch.float32, device=hidden.device)
    lse = torch.empty(rows, dtype=torch.float32, device=hidden.device)
    FLOPS['gemm'] += 2.0 * rows * V * h
    call('aa_lmhead_logprob_fwd', hidden.data_ptr(), hidden.stride(0), w.data_ptr(), w.stride(0), labels.data_ptr(), logp.data_ptr(),
         lse.data_ptr(), ws.data_ptr(), nb, rows, V, h, chunk, dt, int(round_bf16), stream())
    return logp, lse


def lmhead_logprob_bwd(hidden, w, labels, lse, dlogp, dw=None, accumulate=False, chunk=None):
    """Backward of lmhead_logprob_fwd: recomputes each chunk of logits, returns d_hidden [rows, h]; `dw` [V, h] (bf16 or fp32) receives
    (accumulate: is added) the lm_head weight gradient."""
    rows, h = hidden.shape
    V = w.shape[0]
    chunk = int(chunk or LMHEAD_CHUNK)
    dt = 0 if hidden.dtype == bf16 else 1
    nb = _lmhead_ws_bytes(rows, chunk, h, dt, 1)
    ws = torch.empty(nb, dtype=torch.uint8, device=hidden.device)
    d_hidden = torch.empty_like(hidden)
    FLOPS['gemm'] += 2.0 * rows * V * h * (3 if dw is not None else 2)
    call('aa_lmhead_logprob_bwd', hidden.data_ptr(), hidden.stride(0), w.data_ptr(), w.stride(0), labels.data_ptr(), lse.data_ptr(),
         dlogp.data_ptr(), d_hidden.data_ptr(), d_hidden.stride(0), _p(dw), dw.stride(0) if dw is not None else 0,
         int(dw is not None and dw.dtype == torch.float32), int(accumulate), ws.data_ptr(), nb, rows, V, h, chunk, dt, stream())
    return d_hidden


def window_labels(ids, pad_id, resp_len_i32, row_off_i32, labels_out):
    N, T = ids.shape
    call('aa_window_labels', ids.data_ptr(), N, T, int(pad_id), resp_len_i32.data_ptr(), row_off_i32.data_ptr(),
         labels_out.data_ptr(), stream())
    return labels_out


def dpo_loss(pol_logp, ref_logp, seq_off, B, beta, want_grad=True, keep=None):
    dev = pol_logp.device
    out6 = torch.empty(6, dtype=torch.float32, device=dev)
    per = torch.empty((4, B), dtype=torch.float32, device=dev)
    dlogp = torch.zeros_like(pol_logp) if want_grad else None  # pad rows (rows..rows_pad) must be exactly 0
    call('aa_dpo_loss_fwd_bwd', pol_logp.data_ptr(), ref_logp.data_ptr(), seq_off.data_ptr(), int(B), float(beta),
         out6.data_ptr(), per.data_ptr(), _p(dlogp), _p(keep), stream())
    return out6, per, dlogp


PREF_KINDS = {'simpo': 0, 'orpo': 1, 'kto': 2}


def pair_slice_index(ids, mask, seq_off, B):
    """Flat-row slice ranges of the reference's [diverge_index, end_index + 1) window slicing (simpo.py:64-77)."""
    N, T = ids.shape
    dev = ids.device
    lo = torch.empty(N, dtype=torch.int32, device=dev)
    hi = torch.empty(N, dtype=torch.int32, device=dev)
    ln = torch.empty(N, dtype=torch.int32, device=dev)
    keep = torch.empty(B, dtype=torch.uint8, device=dev)
    ids = ids.contiguous(); mask = mask.to(torch.int64).contiguous()
    call('aa_pair_slice_index', ids.data_ptr(), mask.data_ptr(), int(B), T, seq_off.data_ptr(), lo.data_ptr(), hi.data_ptr(),
         ln.data_ptr(), keep.data_ptr(), stream())
    return lo, hi, ln, keep


def sft_loss(logp, rows, want_grad=True):
    """-mean(logp[:rows]) and its gradient w.r.t. the flat window log-probs (pad rows zero)."""
    loss = torch.empty(1, dtype=torch.float32, device=logp.device)
    dlogp = torch.empty_like(logp) if want_grad else None
    call('aa_sft_loss_fwd_bwd', logp.data_ptr(), int(rows), logp.numel(), loss.data_ptr(), _p(dlogp), stream())
    return loss[0], dlogp


def pref_loss(kind, pol_logp, ref_logp, lo, hi, ln, keep, B, beta, p1=0.0, p2=0.0, p3=0.0, want_grad=True):
    dev = pol_logp.device
    out7 = torch.empty(7, dtype=torch.float32, device=dev)
    per = torch.empty((4, B), dtype=torch.float32, device=dev)
    dlogp = torch.empty_like(pol_logp) if want_grad else None     # the kernel zero-fills it
    call('aa_pref_loss_fwd_bwd', PREF_KINDS[kind], pol_logp.data_ptr(), _p(ref_logp), lo.data_ptr(), hi.data_ptr(),
         ln.data_ptr(), keep.data_ptr(), int(B), pol_logp.numel(), float(beta), float(p1), float(p2), float(p3),
         out7.data_ptr(), per.data_ptr(), _p(dlogp), stream())
    return out7, per, dlogp


def window_kl(pol_logp, ref_logp, rows, denom):
    out = torch.empty(1, dtype=torch.float32, device=pol_logp.device)
    call('aa_window_kl', pol_logp.data_ptr(), ref_logp.data_ptr(), int(rows), float(denom), out.data_ptr(), stream())
    return out


def rm_loss(end_scores, B, regularization, want_grad=True):
    out2 = torch.empty(2, dtype=torch.float32, device=end_scores.device)
    d = torch.empty_like(end_scores) if want_grad else None
    call('aa_rm_loss_fwd_bwd', end_scores.data_ptr(), int(B), float(regularization), out2.data_ptr(), _p(d), stream())
    return out2, d


def kl_reward(reward, logp, ref_logp, mask_u8, kl_coeff, clip):
    B, L = logp.shape
    out = torch.empty_like(logp)
    end = torch.empty(B, dtype=torch.int32, device=logp.device)
    call('aa_kl_reward', reward.data_ptr(), logp.data_ptr(), ref_logp.data_ptr(), mask_u8.data_ptr(), B, L,
         float(kl_coeff), float(clip), out.data_ptr(), end.data_ptr(), stream())
    return out, end


def gae(values, rewards, mask_u8, start, gamma, lam):
    B, L = values.shape
    adv = torch.empty((B, L - start), dtype=torch.float32, device=values.device)
    ret = torch.empty_like(adv)
    call('aa_gae', values.data_ptr(), rewards.data_ptr(), mask_u8.data_ptr(), B, L, int(start), float(gamma),
         float(lam), adv.data_ptr(), ret.data_ptr(), stream())
    return adv, ret


def ppo_actor_loss(logp, old_logp, adv, mask_u8, clip_ratio, want_grad=True):
    B, L = logp.shape
    scratch = torch.empty(B, dtype=torch.float32, device=logp.device)
    loss = torch.empty(1, dtype=torch.float32, device=logp.device)
    d = torch.empty_like(logp) if want_grad else None
    call('aa_ppo_actor_loss', logp.data_ptr(), old_logp.data_ptr(), adv.data_ptr(), mask_u8.data_ptr(), B, L,
         float(clip_ratio), scratch.data_ptr(), loss.data_ptr(), _p(d), stream())
    return loss, d


def ppo_critic_loss(values, old_values, returns, mask_u8, clip_value, want_grad=True):
    B, L = values.shape
    scratch = torch.empty(B, dtype=torch.float32, device=values.device)
    loss = torch.empty(1, dtype=torch.float32, device=values.device)
    d = torch.empty_like(values) if want_grad else None
    call('aa_ppo_critic_loss', values.data_ptr(), old_values.data_ptr(), returns.data_ptr(), mask_u8.data_ptr(), B, L,
         float(clip_value), scratch.data_ptr(), loss.data_ptr(), _p(d), stream())
    return loss, d


def group_advantage(rewards, B, G):
    adv = torch.empty_like(rewards)
    call('aa_group_advantage', rewards.data_ptr(), int(B), int(G), adv.data_ptr(), stream())
    return adv


def completion_mask(tokens, eos):
    rows, L = tokens.shape
    mask = torch.empty((rows, L), dtype=torch.uint8, device=tokens.device)
    call('aa_completion_mask', tokens.data_ptr(), tokens.stride(0), rows, L, int(eos), mask.data_ptr(), stream())
    return mask


def grpo_loss(logp, ref_logp, adv, mask_u8, beta, want_grad=True):
    rows, L = logp.shape
    scratch = torch.empty(2 * rows, dtype=torch.float32, device=logp.device)
    loss = torch.empty(1, dtype=torch.float32, device=logp.device)
    d = torch.empty_like(logp) if want_grad else None
    call('aa_grpo_loss_fwd_bwd', logp.data_ptr(), ref_logp.data_ptr(), adv.data_ptr(), mask_u8.data_ptr(), rows, L,
         float(beta), scratch.data_ptr(), loss.data_ptr(), _p(d), stream())
    return loss, d


# ------------------------------------------------------------------ optimizer
SUMSQ_WS = 2048   # AA_SUMSQ_WS


def grad_sumsq_(g, out_accum, scale=1.0, ws=None):
    dt = 0 if g.dtype == bf16 else 1
    if ws is None:
        ws = torch.empty(SUMSQ_WS, dtype=torch.float32, device=g.device)
    tok = _kprof_begin('grad_sumsq')
    call('aa_grad_sumsq', g.data_ptr(), dt, g.numel(), float(scale), out_accum.data_ptr(), ws.data_ptr(), stream())
    _kprof_end(tok, g.element_size() * g.numel())


def chunk_sum(recv, out, world):
    """out[c] = sum over the `world` chunks of recv[world * c] (fp32 accumulation, rank order): reduce step of the direct gradient exchange."""
    dt = 0 if recv.dtype == bf16 else 1
    call('aa_chunk_sum', recv.data_ptr(), out.data_ptr(), dt, out.numel(), int(world), stream())


def clip_coef(sumsq, max_norm, coef_out, norm_out=None):
    call('aa_clip_coef', sumsq.data_ptr(), float(max_norm), coef_out.data_ptr(), _p(norm_out), stream())


def adamw_set_thin(on: bool) -> None:
    """<= 16-VGPR update kernel that can be co-resident with the 256x256 GEMM tiles (overlapped optimizer)."""
    call('aa_adamw_set_thin', int(bool(on)))


def adamw_flat_(master, m, v, p16, g, lr, beta1, beta2, eps, wd, step, gscale=1.0, clip=None):
    dt = 0 if g.dtype == bf16 else 1
    tok = _kprof_begin('adamw_flat')
    call('aa_adamw_flat', master.data_ptr(), m.data_ptr(), v.data_ptr(), _p(p16), g.data_ptr(), dt,
         master.numel(), float(lr), float(beta1), float(beta2), float(eps), float(wd), int(step), float(gscale),
         _p(clip), stream())
    _kprof_end(tok, master.numel() * (24 + g.element_size() + (2 if p16 is not None else 0)))      # 28 B per bf16 parameter


# ------------------------------------------------------------------ decode
# decode step: fold RMSNorm / SwiGLU into the weight streams that consume them (aa_gemm_skinny_fused_bf16).  Measured SLOWER at 7B
# (7.0 vs 5.6 ms per position at 4 sequences: the prologue's VALU work and registers cost the stream more than the two saved
# launches, profiles/r01_bench_decode_7b_v2.json) -> off by default, AA_DECODE_FUSED=1 selects it.
DECODE_FUSED = os.environ.get('AA_DECODE_FUSED', '0') == '1'


class SwizzledWeight:
    """A [N, K] bf16 matrix re-arranged by aa_swizzle_weights_bf16 for the rollout's strip kernel (1 KB contiguous per wave load).
    mode 'glu' / 'rope128' additionally permutes the rows inside every strip so that the strip kernel can finish the element-wise kernel
    that follows the projection in its epilogue (aa_swizzle_weights_perm_bf16: [gate; up] pairs / rotation pairs of head_dim-128 heads)."""
    MODES = {'plain': 0, 'glu': 1, 'rope128': 2}

    def __init__(self, w, mode='plain', kscale=None):
        """kscale: the weight [K] of the RMSNorm in front of this projection -> the copy holds W diag(kscale) (`folded`), and the strip kernel
        takes the un-normalised residual stream and an eps (aa_gemm_skinny_swz_norm_*: the norm's own launch disappears)."""
        if w.dim() != 2 or w.dtype != bf16 or w.stride(1) != 1 or w.shape[1] % 32:
            raise RuntimeError(f'SwizzledWeight: expected a row-major bf16 [N, K] matrix with K % 32 == 0, got {tuple(w.shape)} {w.dtype}')
        self.N, self.K, self.mode = int(w.shape[0]), int(w.shape[1]), mode
        if (mode == 'glu' and self.N % 16) or (mode == 'rope128' and self.N % 128) or mode not in self.MODES:
            raise RuntimeError(f'SwizzledWeight: mode {mode!r} does not fit N = {self.N}')
        if kscale is not None and (kscale.dtype != bf16 or kscale.numel() != self.K or not kscale.is_contiguous()):
            raise RuntimeError(f'SwizzledWeight: kscale must be a contiguous bf16 vector of K = {self.K} elements')
        self.kscale = kscale
        self.data = torch.empty(((self.N + 15) // 16) * 16 * self.K, dtype=bf16, device=w.device)
        self.update(w)

    @property
    def folded(self):
        return self.kscale is not None

    def update(self, w):
        """Re-arrange the current values of `w` (and of the folded norm weight) into the existing storage (the weights moved since the last rollout)."""
        if tuple(w.shape) != (self.N, self.K) or w.dtype != bf16 or w.stride(1) != 1:
            raise RuntimeError(f'SwizzledWeight.update: expected bf16 {(self.N, self.K)}, got {tuple(w.shape)} {w.dtype}')
        if self.kscale is not None:
            call('aa_swizzle_weights_scaled_bf16', w.data_ptr(), w.stride(0), self.data.data_ptr(), self.N, self.K, self.MODES[self.mode], self.kscale.data_ptr(), stream())
        elif self.mode == 'plain':
            call('aa_swizzle_weights_bf16', w.data_ptr(), w.stride(0), self.data.data_ptr(), self.N, self.K, stream())
        else:
            call('aa_swizzle_weights_perm_bf16', w.data_ptr(), w.stride(0), self.data.data_ptr(), self.N, self.K, self.MODES[self.mode], stream())
        return self

    @property
    def shape(self):
        return (self.N, self.K)


def _fold_eps(w, eps, name):
    """A folded copy (W diag(norm weight)) takes the un-normalised rows and the norm's eps; a plain copy must not get one."""
    if w.folded != (eps is not None):
        raise RuntimeError(f'{name}: ' + ('this weight copy has the RMSNorm weight folded in: pass the un-normalised rows and eps' if w.folded
                                          else 'eps given, but the weight copy was made without a norm weight (SwizzledWeight(kscale=...))'))
    return w.folded


def gemm_skinny_glu(x, w, eps=None):
    """act [M, F] = silu(x Wg^T) * (x Wu^T) of a decode position from the 'glu' strip-major copy of the fused [gate; up] weight: the GEMV and
    aa_swiglu_fwd in one launch (bit-identical to the pair).  eps (folded copy): x is the un-normalised residual stream, act = swiglu(rmsnorm(x) W^T)."""
    if not isinstance(w, SwizzledWeight) or w.mode != 'glu' or x.shape[0] > 16 or x.shape[1] != w.K or x.dtype != bf16:
        raise RuntimeError('gemm_skinny_glu: needs a SwizzledWeight(mode="glu"), M <= 16 bf16 rows of width K')
    M, F = x.shape[0], w.N // 2
    act = torch.empty((M, F), dtype=bf16, device=x.device)
    if _fold_eps(w, eps, 'gemm_skinny_glu'):
        call('aa_gemm_skinny_swz_norm_glu_bf16', x.data_ptr(), w.data.data_ptr(), act.data_ptr(), M, F, w.K, x.stride(0), act.stride(0), float(eps), stream())
    else:
        call('aa_gemm_skinny_swz_glu_bf16', x.data_ptr(), w.data.data_ptr(), act.data_ptr(), M, F, w.K, x.stride(0), act.stride(0), stream())
    return act


def gemm_skinny_rope_cache(x, w, bias, H, Hkv, pos, cos_t, sin_t, cache, Tmax, slot, eps=None):
    """q [M, H * 128] (rotated) of a decode position; the rotated k heads and the v heads go straight into cache slot `slot[m]`: the q/k/v GEMV
    and aa_decode_rope_cache in one launch, from the 'rope128' strip-major copy of the fused projection (bit-identical to the pair)."""
    if not isinstance(w, SwizzledWeight) or w.mode != 'rope128' or x.shape[0] > 16 or x.shape[1] != w.K or w.N != (H + 2 * Hkv) * 128:
        raise RuntimeError('gemm_skinny_rope_cache: needs a SwizzledWeight(mode="rope128") of (H + 2 Hkv) * 128 rows, M <= 16 rows of width K')
    if cos_t.dtype != bf16 or x.dtype != bf16 or cache.dtype != bf16 or slot.dtype != torch.int64 or pos.dtype != torch.int32:
        raise RuntimeError('gemm_skinny_rope_cache: bf16 activations / tables / cache, int32 pos, int64 slot')
    M = x.shape[0]
    q = torch.empty((M, H * 128), dtype=bf16, device=x.device)
    if _fold_eps(w, eps, 'gemm_skinny_rope_cache'):       # x = the un-normalised residual stream
        call('aa_gemm_skinny_swz_norm_rope_cache_bf16', x.data_ptr(), w.data.data_ptr(), q.data_ptr(), M, int(H), int(Hkv), w.K, x.stride(0), q.stride(0), _p(bias),
             pos.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), cache.data_ptr(), cache.stride(0), int(Tmax), slot.data_ptr(), float(eps), stream())
        return q
    call('aa_gemm_skinny_swz_rope_cache_bf16', x.data_ptr(), w.data.data_ptr(), q.data_ptr(), M, int(H), int(Hkv), w.K, x.stride(0), q.stride(0), _p(bias),
         pos.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), cache.data_ptr(), cache.stride(0), int(Tmax), slot.data_ptr(), stream())
    return q


def linear_small(x, w, bias=None, residual=None, out=None, norm=None, swiglu=False, fold_eps=None):
    """y = x W^T for a handful of rows (decode): HBM-streaming skinny kernel for M <= 16, tiled GEMM beyond.
    norm = (weight, eps): y = RMSNorm(x) W^T; swiglu: x = [gate | up], y = (silu(gate) * up) W^T -- folded into the weight
    stream for M <= 16 (aa_gemm_skinny_fused_bf16), the separate kernels beyond."""
    M = x.shape[0]
    N = w.shape[0]
    if isinstance(w, SwizzledWeight) and _fold_eps(w, fold_eps, 'linear_small'):
        # fold_eps: w is a strip-major copy of W diag(norm weight) -> y = rmsnorm(x) W^T from the un-normalised rows in one launch
        if M > 16 or norm is not None or swiglu or w.mode != 'plain' or w.K != x.shape[1]:
            raise RuntimeError('linear_small(fold_eps=): a plain folded strip-major copy, M <= 16 rows of width K')
        out = torch.empty((M, N), dtype=bf16, device=x.device) if out is None else out
        call('aa_gemm_skinny_swz_norm_bf16', x.data_ptr(), w.data.data_ptr(), out.data_ptr(), M, N, w.K, x.stride(0), out.stride(0), _p(bias),
             _p(residual), residual.stride(0) if residual is not None else 0, float(fold_eps), stream())
        return out
    if M > 16 or not DECODE_FUSED:
        if norm is not None:
            x = rmsnorm_fwd(x, norm[0], norm[1])[0]
        if swiglu:
            x = swiglu_fwd(x)
        norm, swiglu = None, False
    K = x.shape[1] // 2 if swiglu else x.shape[1]
    if isinstance(w, SwizzledWeight):
        if M > 16 or norm is not None or swiglu or w.mode != 'plain':
            raise RuntimeError('linear_small: plain swizzled weights serve the M <= 16 strip kernel only (glu / rope128 copies have their own entry points)')
        if w.K != K:
            raise RuntimeError(f'linear_small: weight {w.shape} does not match K = {K}')
        out = torch.empty((M, N), dtype=bf16, device=x.device) if out is None else out
        call('aa_gemm_skinny_swz_bf16', x.data_ptr(), w.data.data_ptr(), out.data_ptr(), M, N, K, x.stride(0), out.stride(0), _p(bias),
             _p(residual), residual.stride(0) if residual is not None else 0, stream())
        return out
    if M > 16:
        return gemm(x, w, out=out, bias=bias, residual=residual)
    out = torch.empty((M, N), dtype=bf16, device=x.device) if out is None else out
    ldr = residual.stride(0) if residual is not None else 0
    if norm is not None or swiglu:
        if norm is not None and swiglu:
            raise RuntimeError('linear_small: one prologue at a time')
        if w.shape[1] != K:
            raise RuntimeError(f'linear_small: weight {tuple(w.shape)} does not match K = {K}')
        call('aa_gemm_skinny_fused_bf16', x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, x.stride(0), w.stride(0), out.stride(0),
             _p(bias), _p(residual), ldr, 1 if norm is not None else 2, _p(norm[0]) if norm is not None else None,
             float(norm[1]) if norm is not None else 0.0, stream())
        return out
    call('aa_gemm_skinny_bf16', x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, x.stride(0), w.stride(0),
         out.stride(0), _p(bias), _p(residual), ldr, stream())
    return out


def decode_rope_cache(qkv, H, Hkv, hd, pos, cos_t, sin_t, cache, Tmax, slot):
    """RoPE on the q / k heads of the new token's fused row + write of (k, v) into cache slot `slot[n]` (int64 [N])."""
    N = qkv.shape[0]
    if cos_t.dtype != bf16 or qkv.dtype != bf16 or cache.dtype != bf16 or slot.dtype != torch.int64 or pos.dtype != torch.int32:
        raise RuntimeError('decode_rope_cache: bf16 activations / tables / cache, int32 pos, int64 slot')
    call('aa_decode_rope_cache', qkv.data_ptr(), qkv.stride(0), N, int(H), int(Hkv), int(hd), pos.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(),
         cache.data_ptr(), cache.stride(0), int(Tmax), slot.data_ptr(), stream())
    return qkv


def moe_gemv(x, w3, row_expert, x_div):
    """out[r] = x[r // x_div] @ w3[row_expert[r]]^T: the expert matrices of a few routed rows, streamed once each (decode)."""
    R = row_expert.numel()
    E, N, K = w3.shape
    if x.shape[1] != K or x.dtype != bf16 or w3.dtype != bf16 or not w3.is_contiguous() or row_expert.dtype != torch.int32:
        raise RuntimeError(f'moe_gemv: x {tuple(x.shape)} {x.dtype} / w3 {tuple(w3.shape)} {w3.dtype} / row_expert {row_expert.dtype}')
    out = torch.empty((R, N), dtype=bf16, device=x.device)
    call('aa_moe_gemv_bf16', x.data_ptr(), w3.data_ptr(), out.data_ptr(), R, N, K, x.stride(0), w3.stride(1), out.stride(0),
         row_expert.data_ptr(), w3.stride(0), int(x_div), stream())
    return out


def decode_set_rules(mask: int) -> int:
    """Launch rules of the decode kernels (aa_decode_set_rules: bit 0 = round-6 wave rule + four key steps in flight, bit 1 = pipelined deep strips; 0 = the
    round-5 rules).  Returns the previous mask (A/B runs and the bit-identity tests)."""
    import ctypes
    old = ctypes.c_int(0)
    call('aa_decode_set_rules', int(mask), ctypes.addressof(old))
    return int(old.value)


def attn_decode(q, kcache, vcache, Tmax, start, length, N, H, Hkv, hd, scale):
    out = torch.empty((N, H * hd), dtype=bf16, device=q.device)
    call('aa_attn_decode', q.data_ptr(), q.stride(0), kcache.data_ptr(), vcache.data_ptr(), kcache.stride(0), int(Tmax),
         _p(start), length.data_ptr(), out.data_ptr(), out.stride(0), N, H, Hkv, hd, float(scale), stream())
    return out


def mark_seen_(seen, ids):
    """seen[row, ids[row, j]] = 1 (uint8 [rows, V]); ids [rows, L] int64 with unit inner stride."""
    rows, L = ids.shape
    call('aa_mark_seen', ids.data_ptr(), ids.stride(0), rows, L, seen.data_ptr(), seen.stride(0), seen.shape[1], stream())
    return seen


def move_padding_left(seq, pad_token_id):
    rows, L = seq.shape
    seq = seq.contiguous()
    out = torch.empty_like(seq)
    call('aa_move_padding_left', seq.data_ptr(), seq.stride(0), out.data_ptr(), out.stride(0), rows, L, int(pad_token_id), stream())
    return out


def decode_record(selected, unfinished, out, tslot, nact, pad_token_id, eos_token_id):
    """Bookkeeping of one decode position after the selection kernel (aa_decode_record): returns tok [N] (pad for finished rows), writes it into
    out[n, tslot[n]], counts the step in `nact` if any row was unfinished, clears `unfinished` of rows that emitted eos (eos < 0: none)."""
    N = selected.shape[0]
    if selected.dtype != torch.int64 or unfinished.dtype != torch.bool or out.dtype != torch.int64 or tslot.dtype != torch.int64 or nact.dtype != torch.int64 \
            or out.stride(1) != 1 or not (selected.is_contiguous() and unfinished.is_contiguous() and tslot.is_contiguous()):
        raise RuntimeError('decode_record: int64 selected / out / tslot / nact, bool unfinished, contiguous')
    tok = torch.empty_like(selected)
    call('aa_decode_record', selected.data_ptr(), unfinished.data_ptr(), out.data_ptr(), out.stride(0), tslot.data_ptr(), tok.data_ptr(), nact.data_ptr(), N,
         int(pad_token_id), int(eos_token_id), stream())
    return tok


def decode_tick(tslot, pos, length, step):
    """tslot / pos / length += 1 per row, step += 1 (aa_decode_tick): the end of a decode position."""
    if tslot.dtype != torch.int64 or pos.dtype != torch.int32 or length.dtype != torch.int32 or step.dtype != torch.int64:
        raise RuntimeError('decode_tick: int64 tslot / step, int32 pos / length')
    call('aa_decode_tick', tslot.data_ptr(), pos.data_ptr(), length.data_ptr(), step.data_ptr(), tslot.shape[0], stream())


def argmax_rows(logits, seen=None, repetition_penalty=1.0):
    rows, V = logits.shape
    out = torch.empty(rows, dtype=torch.int64, device=logits.device)
    call('aa_argmax_rows', logits.data_ptr(), logits.stride(0), rows, V, _p(seen), seen.stride(0) if seen is not None else 0,
         float(repetition_penalty), out.data_ptr(), stream())
    return out


def sample_top_p(logits, temperature, top_p, uniform, seen=None, repetition_penalty=1.0, top_k=0):
    """HF's sampling warpers in their order: temperature, top-k (0 / None = off), top-p; draw with `uniform` [rows]."""
    rows, V = logits.shape
    out = torch.empty(rows, dtype=torch.int64, device=logits.device)
    call('aa_sample_top_k_top_p', logits.data_ptr(), logits.stride(0), rows, V, float(temperature), int(top_k or 0), float(top_p),
         uniform.data_ptr(), _p(seen), seen.stride(0) if seen is not None else 0, float(repetition_penalty),
         out.data_ptr(), stream())
    return out
