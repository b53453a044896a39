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

Use a process group of its own (`dist.new_group()`), not the one the gradient all-reduce runs on: collectives of one
communicator are serialised, and the exchange of layer l-1 must not queue behind the 400 MB gradient bucket of layer l.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class ExpertParallel:
    def __init__(self, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('expert parallelism needs an initialised torch.distributed process group')
        self.group = group
        self.size, self.rank = dist.get_world_size(group), dist.get_rank(group)
        # gloo (the CPU / one-GPU test backend) has no device all-to-all: stage through the host.  RCCL exchanges in HBM.
        self.host_staged = dist.get_backend(group) == 'gloo'

    def local_experts(self, num_experts: int) -> tuple[int, int]:
        """(first expert, number of experts) held by this rank: contiguous blocks in rank order."""
        if num_experts % self.size:
            raise ValueError(f'{num_experts} experts do not split over {self.size} ranks')
        n = num_experts // self.size
        return self.rank * n, n

    def _all_to_all(self, out, inp, out_splits=None, in_splits=None):
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
