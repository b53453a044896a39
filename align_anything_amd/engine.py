"""DeepSpeedEngine-shaped native engine for one GPU of a data-parallel job.

The reference drives `deepspeed.initialize(...)` engines (align_anything/trainers/base/supervised_trainer.py:
234-271) and calls `.module`, `.backward(loss)`, `.step()`, `.optimizer.param_groups[0]['lr']`, `.train()`,
`.eval()`, `.tput_timer.update_epoch_count()`, `.gradient_checkpointing_enable()`,
`.save_16bit_model(dir, save_filename=)`, `.save_checkpoint(dir)`, `.load_checkpoint(load_dir=)` on them
(SURVEY.md §8b).  This class offers that surface over the native model:

  * backward(): runs the model's explicit backward (HIP kernels); as each decoder layer finishes, its slice
    of the flat bf16 gradient buffer is all-reduced over RCCL/xGMI on a side HIP stream (one ~400 MB bucket
    per 7B layer -- large buckets suit point-to-point xGMI links), overlapping with the remaining backward.
  * step(): global-L2 clip (device-side coefficient, no host sync) + flat AdamW (csrc/optim.hip) + LR schedule.

ZeRO is not reproduced: at 288 GB/GPU the whole optimizer state of a 7B model fits one device, so pure
data parallelism needs exactly one collective per step (the gradient all-reduce); DeepSpeed's per-layer
parameter all-gathers disappear.
"""
from __future__ import annotations

import math
import os
from types import SimpleNamespace

import torch
import torch.distributed as dist

from . import ops


def cosine_with_warmup(step: int, base_lr: float, warmup: int, total: int) -> float:
    """transformers.get_scheduler('cosine') (supervised_trainer.py:251-257): linear warmup, half-cosine decay."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    prog = (step - warmup) / max(1, total - warmup)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * min(1.0, prog))))


class GradReducer:
    """Bucketed SUM reduction of flat gradient slices on a side stream (backend-agnostic: RCCL on GPU, gloo in the CPU tests).  The 1/world scaling
    is folded into the optimizer kernels (gscale).

    Two exchange forms; `AA_DP_REDUCE` = ring | direct | auto (default):
      ring    one `all_reduce` per bucket: whatever schedule RCCL builds over the xGMI links.
      direct  xGMI is a full mesh of point-to-point links (7 per GPU on an 8-GPU node), so a bucket of n elements is exchanged as
              all-to-all (rank r receives chunk r of every peer: n / w elements over each link, all links at once) -> sum of the w chunks in fp32,
              rank order, one rounding (csrc/optim.hip aa_chunk_sum) -> all-gather of the reduced chunks (again n / w per link, all links at once).
              Same bytes per rank as a ring (2 (w - 1) / w x n) but no multi-hop dependency chain, and a bf16 bucket is rounded once, not w - 1
              times.  Replicas stay bit-identical: every chunk is summed by exactly one rank and broadcast.
      auto    the first bucket of a run times both forms on a 64 MiB scratch buffer (HIP events on the communication stream, max over the ranks so
              that all ranks decide alike), checks that they agree to the gradient dtype's rounding, and keeps the faster one -- no 8-GPU node was
              available while this was written, so the choice is measured where it runs instead of guessed (bench.py reports it as
              multi_gpu.reduce_mode)."""

    def __init__(self, group=None, mode=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.handles = []
        self.comm_stream = None
        # AA_COMM_PROF=1 (bench.py --comm-prof): HIP events on the communication stream around every bucket, so a run on real xGMI
        # reports how long each all-reduce took from "bucket ready" to "reduced" next to the backward it overlaps with
        self.prof = os.environ.get('AA_COMM_PROF', '0') == '1'
        self.records = []          # (bytes, ready event, done event) per bucket since the last report()
        self.mode = (mode or os.environ.get('AA_DP_REDUCE', 'auto')).lower()
        if self.mode not in ('ring', 'direct', 'auto'):
            raise ValueError(f'AA_DP_REDUCE / mode must be ring, direct or auto, got {self.mode!r}')
        self.autotune_report = None
        self._ws = {}              # dtype -> (receive buffer [n], reduced chunk [n / w]) of the direct form, grow-only

    # ---- the two exchange forms (called with the communication stream current on a GPU)
    def _direct(self, flat: torch.Tensor):
        w, n = self.world, flat.numel()
        if n % (8 * w) or flat.data_ptr() % 16:
            return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)      # a ragged slice (never the engine's 64-element-aligned buckets)
        c = n // w
        recv, mine = self._workspace(flat.dtype, flat.device, n)
        recv, mine = recv[:n], mine[:c]
        if flat.is_cuda:
            dist.all_to_all_single(recv, flat, group=self.group)
            ops.chunk_sum(recv, mine, w)
            dist.all_gather_into_tensor(flat, mine, group=self.group)
        else:       # gloo / CPU tensors (tests of the plumbing): bytes travel as uint8 (gloo moves no bf16), the same rank-ordered fp32 sum in torch
            dist.all_to_all_single(recv.view(torch.uint8), flat.view(torch.uint8), group=self.group)
            mine.copy_(recv.view(w, c).float().sum(0).to(flat.dtype))
            parts = [torch.empty_like(mine).view(torch.uint8) for _ in range(w)]
            dist.all_gather(parts, mine.view(torch.uint8), group=self.group)
            flat.view(torch.uint8).copy_(torch.cat(parts))

    def _workspace(self, dtype, device, n):
        """(receive buffer [>= n], reduced chunk [>= n / w]) of the direct form, grow-only.  The one step of that form that can fail on ONE rank alone
        (an allocation), which is why `_autotune` runs it first and lets the ranks agree on it before any rank enters the form's collectives."""
        recv, mine = self._ws.get(dtype, (None, None))
        if recv is None or recv.numel() < n:
            recv = torch.empty(n, dtype=dtype, device=device)
            mine = torch.empty(-(-n // self.world), dtype=dtype, device=device)
            self._ws[dtype] = (recv, mine)
        return recv, mine

    def _exchange(self, flat: torch.Tensor):
        if self.mode == 'auto':
            self._autotune(flat)
        if self.mode == 'direct':
            self._direct(flat)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)

    def _autotune(self, like: torch.Tensor, nbytes: int = 64 << 20, reps: int = 3):
        """Collective (every rank reaches it at its first bucket).  Leaves self.mode = 'ring' or 'direct'."""
        import time
        w = self.world
        n = max(8 * w, (nbytes // like.element_size()) // (8 * w) * (8 * w))
        g = torch.Generator(device='cpu').manual_seed(1234 + dist.get_rank(self.group))
        src = (torch.randn(n, generator=g) * 1e-3).to(like.dtype).to(like.device)
        times, outs = {}, {}
        for mode in ('ring', 'direct'):
            buf = src.clone()
            run = (lambda b: self._direct(b)) if mode == 'direct' else (lambda b: dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group))
            # a form must never be entered by some ranks only.  Two agreements (MIN over the ranks), each BEFORE the next collective of the form:
            # (1) what can fail on one rank alone -- the direct form's workspace allocation -- runs first, outside any collective; (2) the warm-up's own
            # failures are symmetric by construction (a backend without one of the collectives raises on every rank at the same call), and its success is
            # agreed again before the timed repetitions
            def agreed(flag: float) -> bool:
                gt = torch.tensor([flag], device=like.device if like.is_cuda else 'cpu')
                dist.all_reduce(gt, op=dist.ReduceOp.MIN, group=self.group)
                return bool(gt.item())

            good = 1.0
            if mode == 'direct':
                try:
                    self._workspace(like.dtype, like.device, n)
                except RuntimeError as ex:
                    good = 0.0
                    self.autotune_report = {**(self.autotune_report or {}), 'error_' + mode: repr(ex)[:200]}
            if not agreed(good):
                times[mode], outs[mode] = 1e30, None
                continue
            try:
                run(buf)                                   # warm-up (communicator setup) and the value check below
                outs[mode] = buf.clone()
            except RuntimeError as ex:                   # a backend without one of the collectives: the other form is used
                good, outs[mode] = 0.0, None
                self.autotune_report = {**(self.autotune_report or {}), 'error_' + mode: repr(ex)[:200]}
            if not agreed(good):
                ms, outs[mode] = float('inf'), None
            elif like.is_cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run(buf)
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1) / reps
            else:
                t0 = time.perf_counter()
                for _ in range(reps):
                    run(buf)
                ms = (time.perf_counter() - t0) * 1e3 / reps
            t = torch.tensor([ms if ms != float('inf') else 1e30], dtype=torch.float64, device=like.device if like.is_cuda else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            times[mode] = float(t.item())
        # same sums up to the rounding of the gradient dtype (a ring rounds its partial sums): compared in the L2 norm, element-wise cancellation aside
        agree = outs['ring'] is not None and outs['direct'] is not None and float((outs['ring'].double() - outs['direct'].double()).norm()) <= \
            (2.0 ** -6 if like.dtype == torch.bfloat16 else 1e-5) * float(outs['direct'].double().norm()) + 1e-30
        ok = torch.tensor([1.0 if agree else 0.0], device=like.device if like.is_cuda else 'cpu')
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        self.mode = 'direct' if (bool(ok.item()) and times['direct'] < times['ring']) else 'ring'
        bus = lambda ms: 2.0 * (w - 1) / w * n * like.element_size() / max(ms, 1e-9) / 1e6
        self.autotune_report = {**(self.autotune_report or {}), 'chosen': self.mode, 'bytes': n * like.element_size(), 'ring_ms': times['ring'], 'direct_ms': times['direct'],
                                'ring_busbw_GBps': bus(times['ring']), 'direct_busbw_GBps': bus(times['direct']), 'forms_agree': bool(ok.item())}

    def reduce_async(self, flat_slice: torch.Tensor):
        if self.world == 1 or flat_slice.numel() == 0:
            return
        if flat_slice.is_cuda:
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream()
            ev = torch.cuda.Event(enable_timing=self.prof)
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self._exchange(flat_slice)               # stream-ordered on the communication stream; wait() joins the stream
                if self.prof:
                    done = torch.cuda.Event(enable_timing=True)
                    done.record(self.comm_stream)
                    self.records.append((flat_slice.numel() * flat_slice.element_size(), ev, done))
        else:
            self._exchange(flat_slice)

    def wait(self):
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def report(self) -> dict:
        """Profiling mode: per-bucket milliseconds from "gradient slice ready" to "all-reduce finished" (queueing behind earlier
        buckets included) and the bus bandwidth a ring all-reduce of that size implies; clears the records.  Synchronises."""
        if not self.records:
            return {'buckets': 0}
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for _, a, b in self.records]
        nbytes = [n for n, _, _ in self.records]
        span = self.records[0][1].elapsed_time(self.records[-1][2])
        out = {'buckets': len(ms), 'bytes_total': int(sum(nbytes)), 'bucket_ms_sum': sum(ms), 'bucket_ms_max': max(ms),
               'first_ready_to_last_done_ms': span,
               # ring all-reduce moves 2 (w-1)/w of the message over every link
               'busbw_GBps_over_span': 2.0 * (self.world - 1) / self.world * sum(nbytes) / max(span, 1e-6) / 1e6,
               'largest_bucket': {'bytes': int(max(nbytes)), 'ms': ms[nbytes.index(max(nbytes))]}}
        self.records = []
        return out


class NativeEngine:
    def __init__(self, module, *, lr=1e-6, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0,
                 total_steps=1, warmup_steps=0, lr_scheduler_type='cosine', group=None, trainable=True,
                 gradient_accumulation_steps=1):
        self.module = module
        self.trainable = trainable
        self.base_lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(float(b) for b in betas), eps, weight_decay
        self.max_grad_norm = max_grad_norm
        # total_steps None = not known yet (the RL / RM trainers learn it from the dataloader handed to train()): a decaying
        # schedule then refuses to step instead of silently decaying to 0 after the first update
        self.total_steps, self.warmup_steps, self.sched = total_steps, warmup_steps, lr_scheduler_type
        self.global_steps = 0
        self.reducer = GradReducer(group)
        self.world = self.reducer.world
        self.tput_timer = SimpleNamespace(update_epoch_count=lambda: None)
        lr0 = self._lr_at(0)
        self.optimizer = SimpleNamespace(param_groups=[{'lr': lr0, 'weight_decay': weight_decay},
                                                        {'lr': lr0, 'weight_decay': 0.0}])
        self._pending = None
        self.gas = max(1, int(gradient_accumulation_steps))
        self.micro_steps = 0
        # The update runs on the launch stream.  Rounds 1 - 4 enqueued it on a side stream so that it "could overlap" the next batch's reference forward; measured
        # in round 5 (profiles/r05_adam_window.txt) it overlaps nothing on the dense path -- a gemm4 workgroup needs whole SIMD register files and AdamW's
        # workgroups sit on every CU -- so the default path no longer pays a stream switch and two events per step for it.
        # AA_ADAM_ASYNC=1 restores the side stream (it does co-run with small-register kernels: vision towers, the MoE stack's 8-wave grouped GEMMs).
        self.async_optimizer = os.environ.get('AA_ADAM_ASYNC', '0') == '1'
        # optional <= 16-VGPR AdamW kernel that can share CUs with the next step's GEMMs (2 x 248 of the 512 VGPRs per lane are
        # theirs).  Measured: no gain over the default kernel (4.655 / 4.623 vs 4.642 / 4.644 pairs/s) -> off.
        self.thin_optimizer = os.environ.get('AA_ADAM_THIN', '0') == '1'
        self._opt_stream = None
        self._opt_done = None
        ep = getattr(module, 'ep', None)
        if ep is not None and ep.size != self.world:
            raise ValueError(f'expert-parallel degree {ep.size} must equal the data-parallel world size {self.world}')
        if trainable:
            if ep is not None and ep.padded:
                ep.attach_engine()      # before any step or rollout: polls never consume the local capacity flag of an engine-owned exchange
            module.init_training()
            dev = module.device
            self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
            self._coef = torch.ones(1, dtype=torch.float32, device=dev)
            self._gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
            self._sumsq_ws = torch.empty(ops.SUMSQ_WS, dtype=torch.float32, device=dev)
            self._layer_slices = self._compute_layer_slices()

    # ---- DeepSpeedEngine-shaped conveniences
    def train(self, mode=True):
        self.module.train(mode)
        return self

    def eval(self):
        self.module.eval()
        return self

    def gradient_checkpointing_enable(self):
        return None  # nothing is recomputed: activations fit in 288 GB HBM

    def __call__(self, *a, **k):
        raise RuntimeError('NativeEngine is not callable: use trainer.compute_log_probs(engine.module, batch)')

    def set_schedule(self, total_steps: int, warmup_ratio: float | None = None, gradient_accumulation_steps: int | None = None):
        """Length of the LR schedule in OPTIMIZER updates (what the reference hands to transformers.get_scheduler,
        base/rl_trainer.py:190-197, base/supervised_trainer.py:236-257) and, optionally, the accumulation depth.  May be called
        until the first update; the trainers' train() use it once the dataloader length is known.  A RESUMED engine whose schedule length
        was never known (load_checkpoint set global_steps, total_steps still None: ADVICE r4, the RM resume under a cosine schedule) may
        still learn it -- nothing has been evaluated with a different length."""
        if self.global_steps and self.total_steps is not None:
            raise RuntimeError('set_schedule() after the first optimizer update')
        if gradient_accumulation_steps is not None:
            if self.micro_steps % self.gas:
                raise RuntimeError('set_schedule(): cannot change the accumulation depth inside an accumulation window')
            self.gas = max(1, int(gradient_accumulation_steps))
            self.micro_steps = self.global_steps * self.gas          # 0 on a fresh engine; a resumed one stays at the boundary of its last update
        self.total_steps = max(1, int(total_steps))
        if warmup_ratio is not None:
            self.warmup_steps = int(self.total_steps * float(warmup_ratio))
        for pg in self.optimizer.param_groups:
            pg['lr'] = self._lr_at(self.global_steps)

    def _lr_at(self, step):
        if self.sched == 'cosine':
            if self.total_steps is None:
                if step == 0:
                    return 0.0 if self.warmup_steps else self.base_lr
                raise RuntimeError("lr_scheduler_type 'cosine' needs the number of optimizer updates: run the trainer's train() (it "
                                   "derives it from the dataloader like the reference), call engine.set_schedule(total), or set "
                                   "train_cfgs.total_training_steps")
            return cosine_with_warmup(step, self.base_lr, self.warmup_steps, self.total_steps)
        if self.sched == 'constant':
            return self.base_lr
        if self.sched == 'constant_with_warmup':      # transformers get_constant_schedule_with_warmup
            return self.base_lr * min(1.0, step / max(1, self.warmup_steps)) if self.warmup_steps else self.base_lr
        if self.sched == 'linear':                    # transformers get_linear_schedule_with_warmup: up to base_lr, then straight down to 0 at `total`
            if self.total_steps is None:
                if step == 0:
                    return 0.0 if self.warmup_steps else self.base_lr
                raise RuntimeError("lr_scheduler_type 'linear' needs the number of optimizer updates (train(), engine.set_schedule(total) or "
                                   "train_cfgs.total_training_steps)")
            if step < self.warmup_steps:
                return self.base_lr * step / max(1, self.warmup_steps)
            return self.base_lr * max(0.0, (self.total_steps - step) / max(1, self.total_steps - self.warmup_steps))
        raise ValueError(f'lr_scheduler_type {self.sched!r} not supported (cosine, linear, constant, constant_with_warmup)')

    def _compute_layer_slices(self):
        """For every decoder layer: the contiguous range of the flat 'mat' gradient buffer it owns."""
        st = self.module.store
        out = {}
        stack = getattr(self.module, 'stack', None)
        if stack is None or 'mat' not in st.gflat:
            return out
        for L in stack.layers:
            names = [v.wname for v in L.values() if hasattr(v, 'wname')]
            specs = [st.specs[n] for n in names if n in st.specs and st.specs[n]['group'] == 'mat']      # aliases of a fused block (q / k / v views) are not blocks
            if specs:
                lo = min(s['offset'] for s in specs)
                hi = max(s['offset'] + s['numel'] for s in specs)
                out[id(L)] = (lo, hi)
        return out

    # ---- backward / step
    def set_pending(self, dlogp):
        self._pending = dlogp

    def backward(self, loss=None):
        """engine.backward(loss) of the reference (dpo.py:212): seeds come from the fused loss kernel."""
        if self._pending is None:
            raise RuntimeError('backward() without a pending loss gradient: call trainer.loss(batch) first')
        st = self.module.store
        self.wait_optimizer()
        # gradient accumulation (DeepSpeed semantics: backward() scales the loss by 1/gas, step() is a no-op until the
        # boundary): the first micro-batch overwrites / zeroes, later ones accumulate (store.accumulate is read by the
        # dW GEMMs); gradients are exchanged between ranks only at the boundary.
        first = (self.micro_steps % self.gas) == 0
        boundary = ((self.micro_steps + 1) % self.gas) == 0
        st.accumulate = not first
        if first:
            st.zero_grad()
        if self.gas > 1:
            self._pending = self._pending * (1.0 / self.gas)
        done = set()

        def on_layer_done(L):
            rng = self._layer_slices.get(id(L))
            if rng is not None:
                self.reducer.reduce_async(st.gflat['mat'][rng[0]:rng[1]])
                done.add(rng)

        hook = on_layer_done if (self.world > 1 and boundary) else None
        if self.reducer.prof and self.module.device.type == 'cuda':
            self._bwd_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._bwd_ev[0].record()
        self.module.backward_from_dlogp(self._pending, hook)
        if self.reducer.prof and self.module.device.type == 'cuda':
            self._bwd_ev[1].record()
        self._pending = None
        st.accumulate = False
        self.micro_steps += 1
        if self.world > 1 and boundary:
            # whatever was not covered by a per-layer bucket (lm_head, projector, embeddings, vectors)
            mat = st.gflat.get('mat')
            if mat is not None:
                covered = sorted(done)
                pos = 0
                for lo, hi in covered:
                    if lo > pos:
                        self.reducer.reduce_async(mat[pos:lo])
                    pos = max(pos, hi)
                if pos < mat.numel():
                    self.reducer.reduce_async(mat[pos:])
            for g in ('emb', 'vec'):
                if g in st.gflat:
                    self.reducer.reduce_async(st.gflat[g])

    def step(self):
        """Clip + AdamW (three streaming kernels over the flat buffers, csrc/optim.hip) on the launch stream; `async_optimizer` (AA_ADAM_ASYNC=1) moves them
        to a side HIP stream, `wait_optimizer()` is then the join every reader of the weights goes through (a no-op otherwise)."""
        st = self.module.store
        if self.micro_steps % self.gas != 0:
            return  # not at a gradient-accumulation boundary (DeepSpeedEngine.step semantics)
        # HF schedulers are stepped AFTER optimizer.step(): update k (1-based) uses lr(k-1); the value the
        # trainer logs as train/lr after the step is lr(k).  Computed BEFORE any state changes: a schedule that cannot be evaluated
        # yet (cosine / linear without total_steps) raises here and leaves the engine un-stepped, so set_schedule() can still be called
        lr_used = self._lr_at(self.global_steps)
        lr = self._lr_at(self.global_steps + 1)
        self.reducer.wait()
        self.global_steps += 1
        gscale = 1.0 / self.world
        ep = getattr(self.module, 'ep', None)
        watch = ep is not None and ep.padded
        if watch:
            ep.poll_overflow()        # non-blocking: raises (on every rank, in the same step) when an earlier step's shared capacity flag has landed set

        def launch():
            self._sumsq.zero_()
            groups = st.trainable_groups()
            if 'exp' in groups or watch:
                # expert-parallel shards (expert_parallel.py): every rank owns different experts, whose gradients are already the
                # sum over all ranks' rows -> never all-reduced; the clip norm needs the sum of the shards' squared norms.  The same
                # all-reduce carries the capacity-overflow flag of the sync-free exchange (-inf survives the SUM): every rank then skips
                # this update on the device (csrc/optim.hip: coefficient -1) and raises at its next poll -- together.
                if 'exp' in groups:
                    ops.grad_sumsq_(st.gflat['exp'], self._sumsq, gscale, self._sumsq_ws)
                if watch:
                    self._sumsq.add_(ep.overflow_sentinel(self._sumsq.device))
                if self.world > 1:
                    dist.all_reduce(self._sumsq, op=dist.ReduceOp.SUM, group=self.reducer.group)
                if watch:
                    ep.watch_shared(self._sumsq)
            for g in groups:
                if g != 'exp':
                    ops.grad_sumsq_(st.gflat[g], self._sumsq, gscale, self._sumsq_ws)
            ops.clip_coef(self._sumsq, self.max_grad_norm if self.max_grad_norm else 0.0, self._coef, self._gnorm)
            for g in groups:
                wd = 0.0 if g == 'vec' else self.weight_decay
                ops.adamw_flat_(st.master[g], st.m[g], st.v[g], None if st.master[g] is st.flat[g] else st.flat[g], st.gflat[g], lr_used, self.betas[0],
                                self.betas[1], self.eps, wd, self.global_steps, gscale, self._coef)

        if self.async_optimizer and self.module.device.type == 'cuda':
            ops.adamw_set_thin(self.thin_optimizer)
            if self._opt_stream is None:
                self._opt_stream = torch.cuda.Stream()
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self._opt_stream):
                self._opt_stream.wait_event(ev)
                launch()
                self._opt_done = torch.cuda.Event()
                self._opt_done.record(self._opt_stream)
        else:
            ops.adamw_set_thin(False)
            launch()
        for pg in self.optimizer.param_groups:
            pg['lr'] = lr

    def wait_optimizer(self):
        """Make the current stream wait for the last (asynchronous) optimizer update."""
        if self._opt_done is not None:
            torch.cuda.current_stream().wait_event(self._opt_done)
            self._opt_done = None

    def comm_report(self) -> dict:
        """AA_COMM_PROF=1: the last backward's duration next to the gradient buckets it overlapped with (GradReducer.report)."""
        rep = self.reducer.report()
        ev = getattr(self, '_bwd_ev', None)
        if ev is not None:
            torch.cuda.synchronize()
            rep['backward_ms'] = ev[0].elapsed_time(ev[1])
        return rep

    def replica_checksum(self) -> torch.Tensor:
        """Order-sensitive 64-bit checksum of every replicated weight group (bf16 / fp32 bit patterns as integers), one int64 per
        group: pure data parallelism keeps the replicas bit-identical (same all-reduced gradients, deterministic clip norm and
        AdamW), so all ranks must agree on it after any number of steps.  Expert-parallel shards ('exp') differ by design."""
        self.wait_optimizer()
        st = self.module.store
        sums = []
        for g in sorted(st.flat):
            if g == 'exp':
                continue
            t = st.flat[g].view(-1)
            t = t.view(torch.int16 if t.element_size() == 2 else torch.int32)
            acc = torch.zeros((), dtype=torch.int64, device=t.device)
            CH = 1 << 24                                   # bounded temporaries: 3 x 128 MB per chunk
            w = (torch.arange(CH, device=t.device, dtype=torch.int64) % 8191) + 1
            for lo in range(0, t.numel(), CH):
                v = t[lo:lo + CH].to(torch.int64)
                acc += (v * w[:v.numel()]).sum() * (1 + (lo // CH) % 127)
            sums.append(acc)
        return torch.stack(sums)

    def grad_norm(self) -> float:
        self.wait_optimizer()
        ep = getattr(self.module, 'ep', None)
        if ep is not None and ep.padded:
            ep.poll_overflow(block=True)     # a host read anyway: report a capacity overflow of this step now rather than a step later
        return float(self._gnorm.item())     # -1 would be the skipped-update sentinel, which the poll above has already raised on

    def _poll_ep_blocking(self):
        """Before anything is written: a capacity overflow of the sync-free expert exchange (identical flag on every rank) raises here, so a
        checkpoint is never taken of a run that is about to abort."""
        ep = getattr(self.module, 'ep', None)
        if ep is not None and ep.padded:
            ep.poll_overflow(block=True)

    # ---- checkpoints (HF layout, supervised_trainer.py:404-450)
    def save_16bit_model(self, save_dir, save_filename='pytorch_model.bin'):
        """Rank 0 writes (the replicas are identical; DeepSpeed's save_16bit_model writes from rank 0 as well).  Every rank may call it --
        with expert-parallel weights every rank MUST: `state_dict()` gathers the expert shards, a collective."""
        self.wait_optimizer()
        self._poll_ep_blocking()
        path = os.path.join(save_dir, save_filename)
        rank = dist.get_rank(self.reducer.group) if (dist.is_available() and dist.is_initialized()) else 0     # the writer is rank 0 of THIS engine's group
        if rank != 0 and getattr(self.module, 'ep', None) is None:
            return path
        sd = {k: v.cpu() for k, v in self.module.state_dict().items()}
        if rank != 0:
            return path
        os.makedirs(save_dir, exist_ok=True)
        if save_filename.endswith('.safetensors'):
            from safetensors.torch import save_file
            tied = self.module.kind == 'opt' or getattr(self.module, 'tied', False)      # one tensor under two names: safetensors stores it once (HF re-ties on load)
            save_file({k: v.contiguous() for k, v in sd.items() if k != 'lm_head.weight' or not tied}, path, metadata={'format': 'pt'})
        else:
            torch.save(sd, path)
        return path

    def save_checkpoint(self, save_dir, tag=None):
        """Full training state (fp32 masters + Adam moments + step), the analogue of DeepSpeed's checkpoint.  Replicated
        groups are written by rank 0; an expert-parallel shard ('exp') by the rank that owns it (`..._ep<rank>.pt`)."""
        self.wait_optimizer()
        self._poll_ep_blocking()
        os.makedirs(save_dir, exist_ok=True)
        st = self.module.store
        rank = dist.get_rank(self.reducer.group) if dist.is_initialized() else 0
        tag = tag or 'latest'
        pick = lambda groups: {k: {g: t.cpu() for g, t in d.items() if g in groups} for k, d in (('master', st.master), ('m', st.m), ('v', st.v))}
        if rank == 0:
            torch.save({'global_steps': self.global_steps, 'micro_steps': self.micro_steps, 'gas': self.gas, **pick([g for g in st.master if g != 'exp'])},
                       os.path.join(save_dir, f'native_engine_{tag}.pt'))
        if 'exp' in st.master:
            torch.save(pick(['exp']), os.path.join(save_dir, f'native_engine_{tag}_ep{rank}.pt'))

    def load_checkpoint(self, load_dir, tag=None):
        self.wait_optimizer()
        st = self.module.store
        tag = tag or 'latest'
        ck = torch.load(os.path.join(load_dir, f'native_engine_{tag}.pt'), map_location='cpu')
        self.global_steps = ck['global_steps']
        self.micro_steps = ck.get('micro_steps', self.global_steps * self.gas)        # a slice saved inside an accumulation window resumes inside it
        saved_gas = ck.get('gas', self.gas)
        if saved_gas != self.gas:
            # the saved window phase means nothing at another accumulation depth -- restart at the boundary of the last COMPLETED update (a partial
            # window's micro-batches are dropped, they were scaled by 1 / saved_gas) instead of resuming a window built from fewer, wrongly scaled micro-batches
            if self.micro_steps != self.global_steps * saved_gas:
                print(f'[engine] checkpoint saved with gradient_accumulation_steps={saved_gas} inside a window, loaded with {self.gas}: the partial window is dropped', flush=True)
            self.micro_steps = self.global_steps * self.gas
        if self.micro_steps % self.gas:
            # ... with the window's earlier micro-batches lost (DeepSpeed's checkpoint does not carry the accumulation buffers either): the next backward
            # ACCUMULATES (it is not the window's first), so whatever this engine's gradient buffers held must not leak into the resumed window
            for t in st.gflat.values():
                t.zero_()
        if not (self.total_steps is None and self.sched in ('cosine', 'linear')):      # else: set_schedule() refreshes it once the length is known
            for pg in self.optimizer.param_groups:
                pg['lr'] = self._lr_at(self.global_steps)
        if 'exp' in st.master:
            rank = dist.get_rank(self.reducer.group) if dist.is_initialized() else 0
            shard = torch.load(os.path.join(load_dir, f'native_engine_{tag}_ep{rank}.pt'), map_location='cpu')
            for k in ('master', 'm', 'v'):
                ck[k]['exp'] = shard[k]['exp']
        for g in st.master:
            st.master[g].copy_(ck['master'][g])
            st.m[g].copy_(ck['m'][g])
            st.v[g].copy_(ck['v'][g])
            if st.master[g] is not st.flat[g]:
                ops.f32_to_bf16(st.master[g], out=st.flat[g])
        return load_dir, {}
