"""Expert parallelism for the sparse-MoE backbone ("EP all-to-all is new design, no reference
counterpart" -- the reference trains Qwen3-MoE as plain ZeRO data parallelism, models/qwen3_moe.py:28-60).

Why: Qwen3-30B-A3B has 29.0 B of its 30.5 B parameters in the experts.  A pure-DP replica with full AdamW state needs
16 B/param = 488 GB > 288 GB of HBM; with the experts split over 8 ranks (16 of the 128 per GPU) the policy is
(1.5 + 29.0 / 8) x 16 B = 82 GB and the frozen reference 10 GB.  Attention, router, norms and embeddings stay replicated
(data parallel, gradient all-reduce as before); only routed token rows travel.

Per MoE block and direction, each rank
  1. orders its routed (token, choice) rows expert-major WITHOUT padding (`aa_moe_plan`, align 1) -- experts are contiguous
     per owning rank, so this is the send buffer;
  2. exchanges the per-expert row counts (one small all-to-all + ONE host read: the split sizes of step 3);
  3. all-to-all's the rows to the ranks that own their experts (xGMI is point-to-point: one message per peer, sized by the
     router), runs the local experts through the same 128-row-tile grouped GEMMs as the single-GPU path, and
  4. returns the outputs with the mirrored all-to-all; the sender combines them with the routing weights.
Backward is the same exchange with gradients.  Expert-weight gradients are complete on the owning rank (every rank's rows
for that expert arrived there), so they are never all-reduced; the engine only all-reduces their squared norm for clipping.

Volume per rank and block: rows x k x hidden x 2 B per exchange (268 MB at 8192 tokens, top-8, h = 2048), four exchanges
per layer per step = 1.07 GB, against the 1.21 GB per-layer expert-gradient all-reduce (about 2.1 GB on the wire per rank in a
ring of 8) that expert sharding removes.

Sync-free form (round 3; `capacity_factor`, the trainers' default): step 2's host read -- one device->host sync per MoE block and direction, 96+ per
step at 48 layers -- goes away when every (source, destination) pair exchanges a FIXED number of rows, C = ceil(capacity_factor x rows x k / ranks)
(all rows x k when that is a handful: decode positions), the unused tail zero: the split sizes are constants, the per-expert counts travel in a
device all-to-all and stay on the device, the receiver derives the local expert of every arriving row (or -1: no token) from them with a
searchsorted, and `aa_moe_plan` / the grouped GEMM's tile table already skip rows without an expert.

`rows x k` is the MAXIMUM over the ranks (`shared_pairs`): data-parallel ranks pad to their own batch's longest row and rollouts have different prompt
lengths, but the equal-split all-to-all needs ONE block size on every rank.  The ranks agree on it once per forward / per rollout with a host-side
all-reduce of one integer on a gloo side group (`pass_scope`; no device read, ~0.1 ms) -- a rank with fewer rows simply has a longer zero tail.

Overflow: a router that sends one rank more than C rows loses the rows beyond C for that pass (they come back as zeros) and sets a device flag.  The
engine folds the flag into the squared-gradient-norm buffer it all-reduces anyway (`overflow_sentinel`: -inf survives the SUM), so EVERY rank sees it in
the same step: the clip kernel turns it into the coefficient -1, on which the AdamW kernels return without touching weights or moments (the step is
skipped on the device, csrc/optim.hip), and every rank raises together at its next poll / `grad_norm()` / checkpoint save -- nobody is left waiting in a
collective, and no invalid gradient reaches the weights.  The remedy is a larger `train_cfgs.expert_parallel_capacity_factor` (`ranks` can never
overflow; 0 = the exact exchange above).  Without an engine (rollout only) `poll_overflow` reports the local flag.  The padded layout keeps the
expert-major order inside every rank block, so the results are bit-identical to the exact exchange (tests/test_ep_gloo.py on CPU; tests/test_ep_gpu.py
on hardware).

Use a process group of its own (`dist.new_group()`), not the one the gradient all-reduce runs on: collectives of one
communicator are serialised, and the exchange of layer l-1 must not queue behind the 400 MB gradient bucket of layer l.
"""
from __future__ import annotations

import contextlib

import torch
import torch.distributed as dist


class ExpertParallel:
    def __init__(self, group=None, capacity_factor: float | None = None, dense_below: int = 4096):
        """capacity_factor: None / 0 = exact exchange (one host read per block); f >= 1 = the sync-free capacity-padded exchange with
        C = ceil(f x rows x k / ranks) rows per peer; blocks with rows x k <= dense_below use C = rows x k (cannot overflow, and tiny)."""
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('expert parallelism needs an initialised torch.distributed process group')
        self.group = group
        self.size, self.rank = dist.get_world_size(group), dist.get_rank(group)
        # gloo (the CPU / one-GPU test backend) has no device all-to-all: stage through the host.  RCCL exchanges in HBM.
        self.host_staged = dist.get_backend(group) == 'gloo'
        if capacity_factor is not None and 0 < float(capacity_factor) < 1:
            raise ValueError(f'expert_parallel capacity factor {capacity_factor} < 1 would drop rows on a perfectly balanced router')
        self.capacity_factor = float(capacity_factor) if capacity_factor else None
        self.dense_below = int(dense_below)
        self._overflow = None          # 0-d bool on the device: some rank block of this step was larger than its capacity
        self._pending = None           # (pinned host copy, event) of the previous step's flag
        self._shared = None            # 0-d bool: the all-reduced flag of an engine step (watch_shared)
        self._engine = False           # an engine folds the local flag into its all-reduce: polls then read only the shared flag
        # host integers (block sizes) are agreed on a gloo group: an RCCL all-reduce would need a device tensor and a blocking read.
        # Collective over the members of `group` (like the constructor's caller `dist.new_group()`); only the padded exchange needs it.
        self.host_group = group
        if self.padded and self.size > 1 and not self.host_staged:
            self.host_group = dist.new_group(ranks=dist.get_process_group_ranks(group if group is not None else dist.group.WORLD), backend='gloo')
        self._scope = None             # (local pairs, agreed pairs) inside a pass_scope

    def local_experts(self, num_experts: int) -> tuple[int, int]:
        """(first expert, number of experts) held by this rank: contiguous blocks in rank order."""
        if num_experts % self.size:
            raise ValueError(f'{num_experts} experts do not split over {self.size} ranks')
        n = num_experts // self.size
        return self.rank * n, n

    def _all_to_all(self, out, inp, out_splits=None, in_splits=None):
        if self.size == 1:                # a communicator of one: the exchange is a copy (tests/test_qwen3moe_gpu.py runs the whole block this way under
            out.copy_(inp)                # torch's sync-debug mode to show that the capacity-padded form never reads the device from the host)
            return out
        if self.host_staged and inp.is_cuda:
            # byte views: the host path is dtype-agnostic (bf16 rows travel as raw 16-bit words)
            h_in = inp.contiguous().cpu()
            h_out = torch.empty(out.shape, dtype=out.dtype)
            if h_in.dim() == 2:
                dist.all_to_all_single(h_out.view(torch.uint8), h_in.view(torch.uint8), out_splits, in_splits, group=self.group)
            else:
                dist.all_to_all_single(h_out, h_in, out_splits, in_splits, group=self.group)
            out.copy_(h_out)
        else:
            dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=self.group)
        return out

    def exchange_counts(self, counts: torch.Tensor):
        """counts int32 [E]: rows this rank routes to every expert.  Returns (rows sent to each rank, rows received from each
        rank, recv_counts int64 [size, E_local] on the host: rows rank s sends for each of my experts)."""
        n = counts.numel() // self.size
        recv = torch.empty_like(counts)
        self._all_to_all(recv, counts)
        both = torch.stack([counts, recv]).cpu().to(torch.int64)        # the one host read of the block
        send_c, recv_c = both[0].view(self.size, n), both[1].view(self.size, n)
        return send_c.sum(1).tolist(), recv_c.sum(1).tolist(), recv_c

    def exchange_rows(self, x: torch.Tensor, in_splits, out_splits) -> torch.Tensor:
        """x [sum(in_splits), h] ordered by destination rank -> [sum(out_splits), h] ordered by source rank."""
        if x.shape[0] != sum(in_splits):
            raise RuntimeError(f'exchange_rows: {x.shape[0]} rows but the splits sum to {sum(in_splits)}')
        out = torch.empty((sum(out_splits), x.shape[1]), dtype=x.dtype, device=x.device)
        return self._all_to_all(out, x, list(out_splits), list(in_splits))

    # ---- sync-free capacity-padded exchange: every tensor below stays on the device
    @property
    def padded(self) -> bool:
        return self.capacity_factor is not None

    def shared_pairs(self, pairs: int) -> int:
        """max over the ranks of `pairs` (this rank's rows x k of the current pass): the value `capacity` must be derived from, because the
        equal-split all-to-all needs the same block size everywhere while the ranks' padded batch shapes differ.  One host-side all-reduce of
        an integer (gloo side group) -- once per forward / rollout inside a `pass_scope`, else per call.  Every rank must call it in step."""
        if self.size == 1:
            return int(pairs)
        if self._scope is not None and self._scope[0] == int(pairs):
            return self._scope[1]
        t = torch.tensor([int(pairs)], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.host_group)
        return int(t[0])

    @contextlib.contextmanager
    def pass_scope(self, pairs: int):
        """All MoE blocks of one forward (or all decode positions of one rollout) exchange the same number of pairs: agree once."""
        prev = self._scope
        self._scope = (int(pairs), self.shared_pairs(pairs)) if self.padded else None
        try:
            yield
        finally:
            self._scope = prev

    def capacity(self, pairs: int) -> int:
        """Rows per (source, destination) block for a batch of `pairs` = rows x k routed pairs -- the SAME number on every rank
        (`shared_pairs` of the local counts; modeling.Qwen3MoeStack._ep_experts_fwd)."""
        if pairs <= self.dense_below:
            return max(pairs, 1)
        return min(pairs, -(-int(self.capacity_factor * pairs) // self.size))

    def padded_send_layout(self, counts, src, pos, idx, cap):
        """From the dense expert-major plan of the local pairs (`aa_moe_plan`, align 1: counts [E], src [pairs] = token of every dense row,
        pos [rows, k] = dense row of every pair, idx [rows, k] = its expert) to the padded send buffer [size, cap]:
        returns (send_src int32 [size * cap]: token row to gather, -1 for the zero tail; pos_padded int32 [rows, k]: row of every pair in that
        buffer -- and in the buffer that comes back --, -1 for a pair that did not fit, which also raises the overflow flag)."""
        size, n = self.size, counts.numel() // self.size
        pairs = pos.numel()
        cnt_r = counts.view(size, n).sum(1, dtype=torch.int64)                 # rows for every rank
        off_r = torch.cumsum(cnt_r, 0) - cnt_r                                 # where that rank's block starts in the dense order
        j = torch.arange(cap, device=counts.device)
        dense = (off_r[:, None] + j[None, :]).clamp_(max=max(pairs - 1, 0))
        send_src = torch.where(j[None, :] < cnt_r[:, None], src[:pairs].long()[dense], -1).to(torch.int32).reshape(-1)
        rank_of = torch.div(idx.long(), n, rounding_mode='floor')
        within = pos.long() - off_r[rank_of]
        pos_padded = torch.where(within < cap, rank_of * cap + within, -1).to(torch.int32)
        over = (cnt_r > cap).any()
        self._overflow = over if self._overflow is None else (self._overflow | over)
        return send_src, pos_padded

    def exchange_counts_device(self, counts: torch.Tensor) -> torch.Tensor:
        """counts int32 [E] -> int32 [size, E_local]: rows rank s holds for each of my experts.  No host read."""
        recv = torch.empty_like(counts)
        self._all_to_all(recv, counts)
        return recv.view(self.size, -1)

    def padded_recv_ids(self, recv_counts: torch.Tensor, cap: int) -> torch.Tensor:
        """int32 [size * cap, 1]: local expert of every row of the received buffer (source-rank major, expert-major inside a block), -1 for the
        tail of a block.  A block whose sender overflowed holds that sender's first `cap` rows; the sender's flag reports it."""
        size, n = recv_counts.shape
        cum = recv_counts.long().cumsum(1)
        j = torch.arange(cap, device=recv_counts.device).expand(size, cap).contiguous()
        ids = torch.searchsorted(cum, j, right=True)                           # number of experts whose rows end at or before j
        return torch.where(ids >= n, -1, ids).to(torch.int32).view(-1, 1)

    def exchange_fixed(self, x: torch.Tensor) -> torch.Tensor:
        """x [size * cap, h], block d for rank d -> [size * cap, h], block s from rank s (equal constant splits)."""
        if x.shape[0] % self.size:
            raise RuntimeError(f'exchange_fixed: {x.shape[0]} rows do not split over {self.size} ranks')
        return self._all_to_all(torch.empty_like(x), x)

    def overflow_sentinel(self, device) -> torch.Tensor:
        """0-d float32 for the engine's squared-norm buffer: -inf when a block of this step overflowed on THIS rank, else 0; consumes the local
        flag.  Added before the buffer's SUM all-reduce, the -inf reaches every rank (engine.NativeEngine.step)."""
        flag, self._overflow = self._overflow, None
        if flag is None:
            return torch.zeros((), dtype=torch.float32, device=device)
        return torch.where(flag.to(device), float('-inf'), 0.0).to(torch.float32)

    def attach_engine(self) -> None:
        """A TRAINABLE engine owns the local flag from its construction on (ownership decided by the first `watch_shared` left the engine's
        very first step -- and any rollouts before it -- on the stand-alone branch, whose poll consumed the flag before the sentinel could carry it):
        the flag is consumed only by `overflow_sentinel`, so it always reaches the all-reduce, and `poll_overflow` reads only the shared one."""
        self._engine = True

    def watch_shared(self, sumsq: torch.Tensor) -> None:
        """After the all-reduce: start the asynchronous host copy of "some rank overflowed" (identical on every rank) for `poll_overflow`."""
        self._engine = True
        flag = (sumsq.reshape(-1)[0] == float('-inf'))
        self._shared = flag if self._shared is None else (self._shared | flag)
        if self._pending is None:
            self._start_copy('_shared')

    def _start_copy(self, which='_overflow'):
        flag = getattr(self, which)
        setattr(self, which, None)
        if flag.is_cuda:
            host = torch.empty((), dtype=torch.bool).pin_memory()
            host.copy_(flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending = (host, ev)
        else:
            self._pending = (flag, None)

    def poll_overflow(self, block: bool = False) -> None:
        """Never blocks unless asked to: raises if a flag whose host copy has landed was set, then starts the asynchronous read of the flag
        accumulated since.  With an engine (`watch_shared` has run) the flag is the shared one: same value, same step on every rank, and the
        optimizer update of that step was skipped on the device; this rank's own flag is then left for `overflow_sentinel`.  Without an engine
        (stand-alone rollouts) it is this rank's own."""
        if self._pending is not None:
            host, ev = self._pending
            if block and ev is not None:
                ev.synchronize()
            if ev is None or ev.query():
                self._pending = None
                if bool(host.item()):
                    raise RuntimeError(f'expert-parallel exchange overflowed its capacity (factor {self.capacity_factor}, {self.size} ranks): the router sent one '
                                       f'rank more rows than a block holds and those rows were NOT processed -- the step is invalid (its optimizer update was '
                                       f'skipped on the device).  Raise '
                                       f'train_cfgs.expert_parallel_capacity_factor (<= {self.size} always fits) or set it to 0 for the exact exchange')
        which = '_shared' if self._engine else '_overflow'
        if getattr(self, which) is not None and self._pending is None:
            self._start_copy(which)
            if block:
                self.poll_overflow(block=True)

    def all_gather_rows(self, t: torch.Tensor) -> torch.Tensor:
        """Concatenate equally shaped shards along dim 0 in rank order (checkpoint export of the expert tensors)."""
        if self.host_staged:      # raw bytes: gloo has no bf16
            src = t.contiguous().cpu()
            parts = [torch.empty_like(src) for _ in range(self.size)]
            dist.all_gather([p.view(torch.uint8) for p in parts], src.view(torch.uint8), group=self.group)
        else:
            src = t.contiguous()
            parts = [torch.empty_like(src) for _ in range(self.size)]
            dist.all_gather(parts, src, group=self.group)
        return torch.cat(parts, 0).to(t.device)

    @staticmethod
    def local_expert_ids(recv_counts: torch.Tensor, device) -> torch.Tensor:
        """int32 [rows, 1]: the local expert of every received row (source-rank major, expert minor -- the order the rows
        arrive in)."""
        size, n = recv_counts.shape
        ids = torch.repeat_interleave(torch.arange(n, dtype=torch.int64).repeat(size), recv_counts.reshape(-1))
        return ids.to(torch.int32).view(-1, 1).to(device)
