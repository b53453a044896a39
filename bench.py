"""bench.py -- DPO-step throughput of the MI355X-native hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N=1 default; N>1 spawns its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (also accepted)

A "step" is one full DPO optimizer step on one synthetic micro-batch per GPU, exactly what
DPOTrainer.train_step does (align_anything/trainers/text_to_text/dpo.py:205-237): reference forward + policy
forward on the 2B chosen/rejected rows (CLIP tower included) + fused log-prob/DPO loss + policy backward +
(N>1) bucketed RCCL gradient all-reduce overlapped with backward + global-norm clip + AdamW.  Workload =
BASELINE.json configs[1]: LLaVA-1.5-7B geometry (CLIP-L/14-336 + Llama 32x4096, V=32064), bf16, T=2048 =
BOS + 576 image tokens + text, response R=512, random-init weights, synthetic ids/pixels (no network).

Every step gets its OWN freshly generated batch (ids + pixels resident in HBM before the timed region starts, as the
boundary hands over device tensors): the CLIP tower, the response-window plan and every kernel of the step run inside
the timed region for every step -- nothing is cached across steps, and the loss stays at its random-data level.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant kernel = the bf16 MFMA GEMM, measured with
HIP event pairs around a 1-in-11 sample of the GEMM launches of the timed steps; since round 5 also `power_w_mean` / `sclk_mhz_mean` over exactly the timed region from a
sidecar process, tools/power_sampler.py, and with them `peak_at_sclk`, `frac_of_peak_at_sclk`, `j_per_tflop_step`), `step_mfma` (whole-step MFMA fraction on the FLOPs the step
EXECUTES, with SURVEY.md's algorithmic figure beside it) `cpu_baseline` (the CPU oracle port, bounded sample, live; + the committed figure of the reference's own trainer timed in the build
container), `per_batch` (the same step at 1 and 2 pairs per GPU) and `glu_bwd_plan` (the per-box choice between the fused and the
unfused SwiGLU-backward, csrc/gemm.hip).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

# dmabuf IPC: the host driver of the GPU boxes supports no other, and RCCL / CUDA-tensor sharing across the ranks fails without it (hipIpcGetMemHandle: invalid
# argument).  The driver's launcher exports it; this covers a bare `python -m torch.distributed.run ... bench.py` too.  Before the HIP runtime is loaded.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
MEASURED_COPY_GBS = 6290.0     # /opt/skills/guides/MI355X_MICROARCH.md: "8.0 TB/s spec; 6.29 TB/s measured (float4 copy, 79%)"
TRAFFIC_PROFILES = ('r02_gemm_traffic.json', 'r01_gemm_traffic.json')   # newest first (tools/pmc_traffic.sh writes them)


def flops_per_pair(cfg, T, R, n_img_tok, vision_passes=4):
    """algorithmic FLOPs of one preference pair (causal attention at half, no recompute,
    lm_head over all T positions as the reference executes it, vision tower `vision_passes` times)."""
    t, v = cfg['text'], cfg['vision']
    h, F, L, V = t['hidden_size'], t['intermediate_size'], t['num_layers'], t['vocab_size']
    gemm = 2.0 * T * (L * (4 * h * h + 3 * h * F) + h * V)
    attn = L * 2.0 * T * T * h
    vt = (v['image_size'] // v['patch_size']) ** 2 + 1
    vis = 2.0 * vt * v['num_layers'] * (4 * v['hidden_size'] ** 2 + 2 * v['hidden_size'] * v['intermediate_size']) \
        + v['num_layers'] * 4.0 * vt * vt * v['hidden_size'] + 2.0 * (vt - 1) * 588 * v['hidden_size']
    proj = 2.0 * n_img_tok * (v['hidden_size'] * h + h * h)
    f_seq = gemm + attn + vis + proj
    # policy fwd (2 rows) + ref fwd (2 rows) + policy bwd of LLM + projector (2x fwd)
    total = 4 * (gemm + attn + proj) + vision_passes * vis + 2 * 2 * (gemm + attn) + 2 * 2 * proj
    return total, f_seq


def make_batch(cfg, B, T, R, device, seed):
    """PreferenceCollator layout (datasets/text_image_to_text/preference.py:215-263): rows [0,B) chosen,
    [B,2B) rejected, one image per pair shared by both rows, no padding (fixed (image_patches, seq_len))."""
    import torch
    g = torch.Generator(device='cpu').manual_seed(seed)
    n_img = (cfg['vision']['image_size'] // cfg['vision']['patch_size']) ** 2
    N = 2 * B
    ids = torch.randint(3, cfg['image_token_id'], (N, T), generator=g)
    ids[:, 0] = 1
    ids[:, 1:1 + n_img] = cfg['image_token_id']
    # prompt (BOS + image + text) is shared by chosen and rejected; responses differ
    ids[B:, :T - R] = ids[:B, :T - R]
    S = cfg['vision']['image_size']
    pix = torch.randn(B, 3, S, S, generator=g)
    return {
        'input_ids': ids.to(device), 'attention_mask': torch.ones(N, T, dtype=torch.long, device=device),
        'pixel_values': torch.cat([pix, pix], 0).to(device),
        # host integers a collator knows anyway (no padding here; the pair shares its prompt): read by the opt-in shared-prompt packing, ignored otherwise
        'meta_info': {'response_lens': [R] * N, 'seq_lens': [T] * N, 'shared_prefix_lens': [T - R] * B},
    }


def random_init_(model, seed, std=0.02):
    """Random-init weights of the named architecture (no checkpoints offline): N(0, std) matrices and
    embeddings, norm weights 1, biases 0 -- directly on the device."""
    import torch
    g = torch.Generator(device=model.device).manual_seed(seed)
    st = model.store
    for name, s in st.specs.items():
        p = st.p[name]
        if len(s['shape']) >= 2:
            p.normal_(0.0, std, generator=g)
        elif 'norm' in name and name.endswith('weight'):
            p.fill_(1.0)
        else:
            p.zero_()
    if 'model.vision_tower.embeddings.patch_embedding.weight' in st.p:
        st.p['model.vision_tower.embeddings.patch_embedding.weight'][:, 588:].zero_()


def cpu_baseline(cfg, T, reps=2):
    """The CPU oracle (fp32 torch-CPU port of the reference path, oracle/models.py) on a bounded sample: ONE Llama decoder
    layer of the 7B geometry, forward + backward over the 2 rows of one pair at T=2048, one untimed warm-up (thread pool,
    allocator) then `reps` timed repetitions (median), extrapolated linearly to the per-pair cost (32 layers; policy
    fwd+bwd = 3 forward units, reference fwd = 1: x 4/3).  lm_head, vision tower and optimizer are NOT included, which
    flatters the CPU.  kind = "port": SURVEY.md §8(d) asks for the reference's own DPOTrainer.loss on HF modules, but
    /root/reference (and a DeepSpeed install) do not exist on the GPU box, so the restatement the parity tests pin to the
    reference's fixtures is what can be timed there."""
    import torch
    from oracle import models as om
    t = dict(cfg['text'])
    t['num_layers'] = 1
    torch.manual_seed(0)
    h, F, H, hd = t['hidden_size'], t['intermediate_size'], t['num_heads'], t['head_dim']
    p = 'model.language_model.'
    sd = {p + 'layers.0.input_layernorm.weight': torch.ones(h), p + 'layers.0.post_attention_layernorm.weight': torch.ones(h),
          p + 'norm.weight': torch.ones(h)}
    for n, shape in (('self_attn.q_proj', (H * hd, h)), ('self_attn.k_proj', (H * hd, h)), ('self_attn.v_proj', (H * hd, h)),
                     ('self_attn.o_proj', (h, H * hd)), ('mlp.gate_proj', (F, h)), ('mlp.up_proj', (F, h)), ('mlp.down_proj', (h, F))):
        sd[p + f'layers.0.{n}.weight'] = (torch.randn(shape) * 0.02).requires_grad_(True)
    x = torch.randn(2, T, h) * 0.1
    cores = torch.get_num_threads()

    def once():
        for v in sd.values():
            v.grad = None
        t0 = time.time()
        out = om.llama_decoder(sd, t, x, None)
        out.float().pow(2).mean().backward()
        return time.time() - t0

    warm = once()
    times = sorted(once() for _ in range(max(1, reps)))
    dt = times[len(times) // 2]
    per_pair = dt * cfg['text']['num_layers'] * 4.0 / 3.0
    return {'value': 1.0 / per_pair, 'unit': 'pairs/s', 'cores': cores, 'kind': 'port',
            'sample': f'oracle/models.py llama_decoder, 1 of {cfg["text"]["num_layers"]} layers, fwd+bwd fp32, 2x{T} tokens: warm-up '
                      f'{warm:.1f} s, then {len(times)} timed reps {[round(x, 2) for x in times]} s (median used); extrapolated '
                      f'x{cfg["text"]["num_layers"]} layers x4/3 (policy fwd+bwd + ref fwd); lm_head/vision/optimizer excluded',
            'why_port': 'the reference (HF + DeepSpeed trainer under /root/reference) is not present on the GPU box; the oracle '
                        'port is pinned to reference-generated fixtures by tests/test_oracle_golden.py'}


def preflight(device, rank, world, layers, nbytes=1 << 30, reps=5):
    """Before any model is built (the first real multi-GPU run must not die half-way for a reason a 5-second check finds):
    per rank the visible device count and the free HBM against what one data-parallel replica needs (policy bf16 + fp32
    master / m / v + bf16 gradients + frozen reference + activations of 4 pairs ~ 122 GB + 40 GB at full depth), then one 1 GB bf16 all-reduce
    on RCCL, warmed once and timed `reps` times with HIP events on the collective's stream -- the bus bandwidth (2 (w-1)/w x bytes / time) is what the
    per-layer 405 MB gradient buckets will see (DESIGN.md section 6 budgets against >= 250 GB/s).  Rank 0 returns the gathered report."""
    import torch
    import torch.distributed as dist
    free, total = torch.cuda.mem_get_info(device)
    need = int((122 + 40) * (1 << 30) * max(layers, 1) / 32)
    rep = {'rank': rank, 'visible_devices': torch.cuda.device_count(), 'device': torch.cuda.get_device_name(device), 'free_GiB': round(free / 2 ** 30, 1),
           'total_GiB': round(total / 2 ** 30, 1), 'replica_needs_GiB': round(need / 2 ** 30, 1), 'fits': bool(free > need)}
    if world > 1:
        buf = torch.ones(nbytes // 2, dtype=torch.bfloat16, device=device)
        dist.all_reduce(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        buf.fill_(1.0 / world)
        e0.record()
        for _ in range(reps):
            dist.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        rep['allreduce_1GiB_ms'] = round(ms, 3)
        rep['allreduce_busbw_GBps'] = round(2.0 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9, 1)
        rep['allreduce_value_ok'] = bool(torch.isfinite(buf.float().sum()))
        del buf
        torch.cuda.empty_cache()
    print(f'[bench preflight] {json.dumps(rep)}', file=sys.stderr, flush=True)
    if not rep['fits']:
        print(f'[bench preflight] rank {rank}: only {rep["free_GiB"]} GiB free, a replica needs ~{rep["replica_needs_GiB"]} GiB -- expect an out-of-memory abort',
              file=sys.stderr, flush=True)
    if world > 1:
        allr = [None] * world
        dist.all_gather_object(allr, rep)
        return allr if rank == 0 else None
    return [rep]


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def self_launch(argv, n: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU
    (the reference launches the same way through `deepspeed --master_port ... --module`, scripts/llava/llava_dpo.sh:35-46)."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    if os.environ.get('AA_BENCH_DRYRUN_LAUNCH') == '1':     # tests: show the command, do not run it
        print(json.dumps({'self_launch': cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL needs it on this driver
    env.setdefault('MASTER_ADDR', '127.0.0.1')
    print(f'[bench] spawning {n} ranks: {" ".join(cmd)}', file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--pairs-per-gpu', type=int, default=int(os.environ.get('AA_BENCH_PAIRS', 4)))
    ap.add_argument('--seq-len', type=int, default=2048)
    ap.add_argument('--response-len', type=int, default=512)
    ap.add_argument('--layers', type=int, default=32, help='LLM depth (32 = LLaVA-1.5-7B; anything else is NOT the headline config)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-gemm-events', action='store_true')
    ap.add_argument('--share-prompt', action='store_true',
                    help='run the TIMED steps with shared-prompt packing (train_cfgs.share_prompt_prefix: a pair\'s common prefix once per model).  Off by default: the '
                         'headline keeps the reference\'s row-per-sequence layout; the packed step is always measured after the timed region and reported as `shared_prompt`')
    ap.add_argument('--gemm-event-stride', type=int, default=11,
                    help='time every k-th GEMM launch of the timed steps with a HIP event pair (1 = all).  An event pair isolates its launch from '
                         'its neighbours (no tail / head overlap with the next kernel): around EVERY launch that costs the one-wave-per-SIMD '
                         'GEMMs 6.5 %% of the step (806 vs 754 ms, same box), so the default samples every 11th launch -- made coprime to the '
                         'launches per step, hence unbiased over the shapes -- which costs < 1 %%')
    ap.add_argument('--traffic', choices=('auto', 'measure', 'committed'), default=os.environ.get('AA_BENCH_TRAFFIC', 'auto'),
                    help='where roofline.traffic comes from.  measure: before the timed run, tools/pmc_traffic.sh collects it for THIS box and build '
                         '(two rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 1` at full depth, ~3 min); committed: read profiles/r02_gemm_traffic.json; '
                         'auto (default): measure at N=1 when rocprofv3 is on PATH, else committed.  roofline.traffic_source says which one the line carries')
    ap.add_argument('--measure-traffic', action='store_true', help='same as --traffic measure (kept for the round-2 command lines)')
    ap.add_argument('--no-power', action='store_true', help='do not start the power / clock sidecar (tools/power_sampler.py)')
    ap.add_argument('--no-per-batch', action='store_true', help='skip the B=1 / B=2 pairs-per-GPU datapoints measured after the timed region (N=1 only)')
    ap.add_argument('--rccl-channels', type=int, default=int(os.environ.get('AA_RCCL_CHANNELS', 0)),
                    help='N>1: cap RCCL at this many channels (NCCL_MAX_NCHANNELS; one channel = one workgroup = one CU taken from the GEMMs while a '
                         'gradient bucket is in flight -- a gemm4 workgroup holds all 512 registers of its CU\'s SIMDs, so RCCL and GEMM tiles never share '
                         'a CU; DESIGN.md section 6 has the expected cost).  0 = RCCL\'s default')
    ap.add_argument('--comm-prof', dest='comm_prof', action='store_true', default=None,
                    help='N>1 (default ON there since round 4: the first multi-GPU record must carry it): after the timed region run ONE extra untimed '
                         'step with HIP events around every gradient bucket (all-reduce time vs the backward it overlaps with) -> "multi_gpu.comm"')
    ap.add_argument('--no-comm-prof', dest='comm_prof', action='store_false')
    ap.add_argument('--preflight', dest='preflight', action='store_true', default=None,
                    help='before any model is built, per rank: visible devices, free HBM against the ~122 GB replica, and (N>1) a 1 GB RCCL all-reduce '
                         'timed with HIP events -> bus GB/s on stderr and in "multi_gpu.preflight".  Default: on for N>1, off for N=1')
    ap.add_argument('--no-preflight', dest='preflight', action='store_false')
    ap.add_argument('--preflight-only', action='store_true', help='run the preflight, print its JSON line and exit (no model, no steps)')
    args = ap.parse_args()
    if args.comm_prof is None:
        args.comm_prof = args.gpus > 1
    if args.preflight is None:
        args.preflight = args.gpus > 1 or args.preflight_only

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    live_traffic = None
    in_pmc_child = os.environ.get('AA_BENCH_IN_PMC') == '1'      # this process IS one of the rocprofv3 --pmc passes: no extras
    import shutil
    want_measure = args.measure_traffic or args.traffic == 'measure' or (args.traffic == 'auto' and shutil.which('rocprofv3') is not None)
    if want_measure and args.gpus == 1 and args.layers == 32 and not in_pmc_child:
        env = dict(os.environ, GRAFT_REPO_ROOT=ROOT, AA_BENCH_IN_PMC='1')
        try:
            os.remove(os.path.join(ROOT, 'gpurun_out', 'gemm_traffic.json'))
        except OSError:
            pass
        t_pmc = time.time()
        try:
            r = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'pmc_traffic.sh')], env=env, capture_output=True, text=True,
                               timeout=float(os.environ.get('AA_BENCH_TRAFFIC_TIMEOUT', 900)))
            with open(os.path.join(ROOT, 'gpurun_out', 'gemm_traffic.json')) as f:
                live_traffic = json.load(f)
            if not live_traffic.get('gemm4_launches'):
                raise ValueError('no gemm4 launches in the counter files')
            live_traffic['_source'] = f'measured by this invocation (tools/pmc_traffic.sh, {time.time() - t_pmc:.0f} s before the timed run)'
        except (OSError, ValueError, subprocess.TimeoutExpired) as ex:
            live_traffic = None
            print(f'[bench] in-run traffic measurement failed ({ex!r}); falling back to the committed profile', file=sys.stderr, flush=True)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    # AA_BENCH_ONE_DEVICE=1 (+ AA_BENCH_BACKEND=gloo) is a FUNCTIONAL check of the N>1 code path on a 1-GPU box:
    # all ranks share cuda:0 and the collectives go through gloo.  Never a performance number.
    one_device = os.environ.get('AA_BENCH_ONE_DEVICE') == '1'
    backend = os.environ.get('AA_BENCH_BACKEND', 'nccl')
    if one_device:
        local = 0
    if not one_device and local >= torch.cuda.device_count():
        raise SystemExit(f'rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible '
                         '(AA_BENCH_ONE_DEVICE=1 AA_BENCH_BACKEND=gloo runs the functional check on one device)')
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.rccl_channels > 0:
            os.environ['NCCL_MAX_NCHANNELS'] = str(args.rccl_channels)
            os.environ['NCCL_MIN_NCHANNELS'] = str(min(args.rccl_channels, int(os.environ.get('NCCL_MIN_NCHANNELS', args.rccl_channels))))
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)      # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)
        # per-rank confirmation of what the collectives run on (stderr; stdout carries the one JSON line)
        probe = torch.ones(1, device=device)
        dist.all_reduce(probe)
        print(f'[bench] rank {rank}/{world} device {torch.cuda.get_device_name(local)} #{local} backend={dist.get_backend()} '
              f'world_seen_by_collective={int(probe.item())}', file=sys.stderr, flush=True)

    pre = preflight(device, rank, world, args.layers) if args.preflight else None
    if args.preflight_only:
        if rank == 0:
            print(json.dumps({'preflight': pre}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from align_anything_amd import configs, ops
    from align_anything_amd.trainers.dpo import DPOTrainer

    cfg = configs.llava_1_5_7b(num_layers=args.layers)
    B, T, R = args.pairs_per_gpu, args.seq_len, args.response_len
    cfgs = {'train_cfgs': {'scale_coeff': 0.1, 'learning_rate': 1e-6, 'lr_warmup_ratio': 0.03, 'weight_decay': 0.0,
                           'adam_betas': [0.9, 0.95], 'lr_scheduler_type': 'cosine',
                           'total_training_steps': args.steps + args.warmup + 32, 'freeze_mm_proj': False,
                           'freeze_language_model': False, 'freeze_vision_tower': True, 'share_prompt_prefix': bool(args.share_prompt)},
            'model_cfgs': {'pad_token_id': cfg['pad_token_id']}}
    tr = DPOTrainer(cfgs, {'gradient_clipping': 1.0}, model_cfg=cfg, device=device)
    random_init_(tr.policy, seed=42)
    # reference = same checkpoint as the policy (dpo.py:89-99 loads both from model_name_or_path)
    tr.reference.load_state_dict(tr.policy.state_dict())
    for g in tr.policy.store.master:
        tr.policy.store.master[g].copy_(tr.policy.store.flat[g])
    # one fresh batch per step (warm-up and timed), all resident in HBM before the clock starts
    n_b = args.warmup + args.steps
    batches = [make_batch(cfg, B, T, R, device, seed=1234 + rank * 100003 + i) for i in range(n_b)]
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    # package power + shader clock over the timed region, from a sidecar PROCESS (tools/power_sampler.py) so that the poller
    # shares no interpreter lock with the launch loop; rank 0 samples its own device
    sampler, sampler_path = None, None
    if rank == 0 and not args.no_power and not in_pmc_child:
        import tempfile
        sampler_path = os.path.join(tempfile.gettempdir(), f'aa_power_{os.getpid()}.jsonl')
        try:
            sampler = subprocess.Popen([sys.executable, os.path.join(ROOT, 'tools', 'power_sampler.py'), '--device', str(local), '--out', sampler_path, '--hz', '20'],
                                       stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except OSError:
            sampler = None

    for i in range(args.warmup):
        if i == args.warmup - 1 and not args.no_gemm_events:
            ops.GEMM_PROF = []                      # count the GEMM launches of one step ...
        tr.train_step(batches[i])
    torch.cuda.synchronize()
    if not args.no_gemm_events:                     # ... and create every event the timed region will record up front
        per_step = len(ops.GEMM_PROF) if ops.GEMM_PROF else 600
        ops.GEMM_PROF = None
        # --gemm-event-stride k: every k-th launch carries an event pair (see the flag's help); k is bumped until it shares no factor
        # with the number of GEMM launches per step, so successive steps sample different launches and every shape gets its share
        import math
        ops.GEMM_PROF_STRIDE = max(1, args.gemm_event_stride)
        while ops.GEMM_PROF_STRIDE > 1 and math.gcd(per_step, ops.GEMM_PROF_STRIDE) != 1:
            ops.GEMM_PROF_STRIDE += 1
        ops.event_pool_fill(2 * (per_step * args.steps // ops.GEMM_PROF_STRIDE + 8) + 64 + 2 * 64 * args.steps)      # + the ~55 sampled non-GEMM launches per step
    barrier()

    gemm_events = None if args.no_gemm_events else []
    ops.GEMM_PROF = gemm_events
    # the kernels below the GEMMs (HBM fraction per memory-bound kernel; the attention kernels against the MFMA peak): event pairs on every
    # 7th launch of each kind, every AdamW / squared-norm launch (ops.KPROF)
    ops.KPROF = None if args.no_gemm_events else {}
    ops._kseq.clear()
    ops.FLOPS['gemm'] = ops.FLOPS['attn'] = 0.0
    torch.cuda.synchronize()
    wall0 = time.time()
    t0 = time.perf_counter()
    losses = []
    for i in range(args.steps):
        last = tr.train_step(batches[args.warmup + i])
        losses.append(round(last['train/loss'], 5))
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    wall1 = time.time()
    ops.GEMM_PROF = None
    kprof, ops.KPROF = ops.KPROF, None
    executed = dict(ops.FLOPS)
    power = None
    if sampler is not None:
        try:
            sampler.stdin.close()                       # the sidecar leaves at its next tick; SIGTERM to exactly this PID if it does not
            sampler.wait(timeout=3)
        except (OSError, subprocess.TimeoutExpired):
            sampler.terminate()
            try:
                sampler.wait(timeout=2)                 # reaped, and done writing, before the jsonl is read
            except subprocess.TimeoutExpired:
                sampler.kill()
                sampler.wait()
        try:
            from tools.power_sampler import summarise
            power = summarise(sampler_path, wall0, wall1, PEAK_BF16_TFLOPS)
        finally:
            try:
                os.remove(sampler_path)
            except OSError:
                pass

    tt = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

    # N>1, outside the timed region: (1) the replicas must still be bit-identical after warmup + steps optimizer updates (pure DP:
    # identical all-reduced gradients, deterministic clip norm and AdamW); (2) optionally one extra step with events on the
    # communication stream, to put numbers on the bucket / backward overlap the first time this runs on real xGMI
    multi = None
    if world > 1:
        mine = tr.model.replica_checksum()
        allc = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allc, mine)
        same = all(bool(torch.equal(c, allc[0])) for c in allc)
        multi = {'backend': dist.get_backend(), 'world': world, 'rccl_max_nchannels': os.environ.get('NCCL_MAX_NCHANNELS'), 'replicas_bit_identical_after_steps': same,
                 'optimizer_updates_checked': tr.model.global_steps, 'preflight': pre,
                 # which form of the gradient exchange ran (engine.GradReducer: RCCL's all-reduce or all-to-all + sum + all-gather over the direct links) and the
                 # in-run measurement that chose it
                 'reduce_mode': tr.model.reducer.mode, 'reduce_autotune': tr.model.reducer.autotune_report}
        if not same:
            print(f'[bench] rank {rank}: REPLICAS DIVERGED: {[c.tolist() for c in allc]}', file=sys.stderr, flush=True)
        if args.comm_prof:
            tr.model.reducer.prof = True
            tr.train_step(make_batch(cfg, B, T, R, device, seed=99 + rank))
            multi['comm'] = tr.model.comm_report()
            tr.model.reducer.prof = False

    # N=1, outside the timed region: the same step at 1 and 2 pairs per GPU (the reference's yaml default is
    # per_device_train_batch_size 1, configs/train/text_image_to_text/dpo.yaml:30; SURVEY section 8(d) allows B in {1, 2, 4})
    per_batch = None
    if world == 1 and not args.no_per_batch and not in_pmc_child:
        per_batch = {}
        ops.GEMM_PROF = None
        for b2 in (1, 2):
            if b2 == B:
                continue
            bb = [make_batch(cfg, b2, T, R, device, seed=777 + 10 * b2 + i) for i in range(4)]
            tr.train_step(bb[0])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(1, 4):
                tr.train_step(bb[i])
            torch.cuda.synchronize()
            d1 = (time.perf_counter() - t1) / 3
            per_batch[f'B{b2}'] = {'pairs_per_gpu_step': b2, 'value': b2 / d1, 'unit': 'pairs/s', 'ms_per_step': d1 * 1e3, 'steps': 3, 'warmup': 1}
            del bb

    # N=1, outside the timed region: the same step with shared-prompt packing (opt-in train_cfgs.share_prompt_prefix; trainers/common.py::build_pack_plan): the
    # pair's common prefix -- 1536 of 2048 positions here -- is computed once per model.  Reported beside the headline, never as it (unless --share-prompt).
    shared_prompt = None
    if world == 1 and not args.no_per_batch and not in_pmc_child:
        was = tr.share_prompt_prefix
        tr.share_prompt_prefix = not was
        ops.GEMM_PROF = None
        f0 = dict(ops.FLOPS)
        bb = [make_batch(cfg, B, T, R, device, seed=555 + i) for i in range(5)]
        tr.train_step(bb[0])
        torch.cuda.synchronize()
        ops.FLOPS['gemm'] = ops.FLOPS['attn'] = 0.0
        t1 = time.perf_counter()
        for i in range(1, 5):
            last_sp = tr.train_step(bb[i])
        torch.cuda.synchronize()
        d1 = (time.perf_counter() - t1) / 4
        plan = bb[1].get('_pack')
        shared_prompt = {'share_prompt_prefix': not was, 'value': B / d1, 'unit': 'pairs/s', 'ms_per_step': d1 * 1e3, 'steps': 4, 'warmup': 1,
                         'token_rows_per_step': plan['rows'] if plan else 2 * B * T, 'token_rows_reference_layout': 2 * B * T,
                         'executed_tflop_per_pair': (ops.FLOPS['gemm'] + ops.FLOPS['attn']) / (4 * B) / 1e12, 'loss_last_step': round(last_sp['train/loss'], 5),
                         'note': 'the same DPO step, the same losses up to rounding (tests/test_pack_gpu.py; against the reference trainer at full depth: '
                                 'tests/test_secondary_geometry_gpu.py); fewer FLOPs are EXECUTED per pair -- utilisation figures must use executed_tflop_per_pair'}
        tr.share_prompt_prefix = was
        ops.FLOPS.update(f0)
        del bb

    if rank == 0:
        n_img = (cfg['vision']['image_size'] // cfg['vision']['patch_size']) ** 2
        fl_pair, _ = flops_per_pair(cfg, T, R, n_img)
        pairs = B * world * args.steps
        value = pairs / dt
        step_s = dt / args.steps
        alg_tflops = fl_pair * B / step_s / 1e12   # per GPU
        exe_pair = (executed['gemm'] + executed['attn']) / (B * args.steps)
        exe_tflops = exe_pair * B / step_s / 1e12
        out = {
            'metric': 'preference-pairs/sec (DPO step, LLaVA-1.5-7B geometry, seq=2048), whole job',
            'value': value, 'unit': 'pairs/s', 'per_gpu': value / world, 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': step_s * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'BASELINE configs[1]: LLaVA-1.5-7B DPO, bf16, 336px/576 patches, seq_len={T}, response={R}, '
                                   f'{B} pairs/GPU/step, CLIP tower + policy+ref fwd, bwd, clip, AdamW; a fresh batch every step'
                                   + (' [shared-prompt packing ON: --share-prompt]' if args.share_prompt else '') + ('' if args.layers == 32 else f' [REDUCED DEPTH {args.layers}]')
                                   + (' [FUNCTIONAL CHECK: ranks share one device, gloo]' if one_device or backend != 'nccl' else ''),
                       'global_batch_pairs': B * world, 'seq_len': T, 'parallelism': f'dp{world}',
                       'shapes': {'hidden': cfg['text']['hidden_size'], 'heads': cfg['text']['num_heads'], 'head_dim': cfg['text']['head_dim'],
                                  'ffn': cfg['text']['intermediate_size'], 'vocab': cfg['text']['vocab_size'], 'layers': args.layers,
                                  'tokens_per_gpu_step': 2 * B * T, 'image_tokens': n_img},
                       'trainable_params': tr.policy.store.num_trainable(), 'losses_timed_steps': losses},
            'n1_equivalent': {'value': value / world, 'unit': 'pairs/s per GPU', 'ms_per_step': step_s * 1e3,
                              'note': 'weak scaling: every GPU runs the N=1 workload; compare with the N=1 line directly'},
            'step_mfma': {'executed_tflop_per_pair': exe_pair / 1e12, 'achieved_tflops_per_gpu': exe_tflops,
                          'frac_of_dense_bf16_peak': exe_tflops / PEAK_BF16_TFLOPS,
                          'executed_split': {'gemm_tflop_per_pair': executed['gemm'] / (B * args.steps) / 1e12,
                                             'attention_tflop_per_pair': executed['attn'] / (B * args.steps) / 1e12},
                          'algorithmic_tflop_per_pair': fl_pair / 1e12, 'algorithmic_frac_of_dense_bf16_peak': alg_tflops / PEAK_BF16_TFLOPS,
                          'note': 'headline = FLOPs counted at the launches of the timed steps (2MNK per GEMM, causal attention at half, '
                                  'backward 2.5x forward); algorithmic = SURVEY.md §8(d) accounting (lm_head over all T, 4 vision passes), '
                                  'larger because lm_head runs on the response rows only, the frozen tower once per image, and (round 6) the last decoder layer after its '
                                  'keys / values on the response rows only -- nothing reads its other rows; log-probs bit-identical, AA_TAIL_PRUNE=0 switches it off'},
        }
        if gemm_events:
            tot_ms, tot_fl = 0.0, 0.0
            for ev0, ev1, fl, _ in gemm_events:
                tot_ms += ops.event_elapsed_ms(ev0, ev1)
                tot_fl += fl
            n = len(gemm_events)
            ach = tot_fl / tot_ms / 1e9
            # the same sample split by launch kind (a kind = equal FLOPs and equal algorithmic bytes: one GEMM shape / epilogue of the step)
            kinds = {}
            for ev0, ev1, fl, by in gemm_events:
                k = kinds.setdefault((fl, by), [0, 0.0])
                k[0] += 1
                k[1] += ops.event_elapsed_ms(ev0, ev1)
            by_kind = [{'tflop': round(fl / 1e12, 4), 'algorithmic_MB': round(by / 1e6, 1), 'sampled_launches': c, 'avg_ms': round(ms / c, 4),
                        'tflops': round(fl / (ms / c) / 1e9, 1)} for (fl, by), (c, ms) in sorted(kinds.items(), key=lambda kv: -kv[1][1])][:12]
            # HBM bytes per GEMM launch from the PMC counters: rocprofv3 cannot run inside this process, so the number comes
            # from the separate --pmc passes of THIS command (tools/pmc_traffic.sh: FETCH_SIZE doubled per the gfx950 note of
            # MI355X_MICROARCH.md, WRITE_SIZE as is, separate passes), committed under profiles/
            traffic, src = None, None
            if live_traffic is not None:
                traffic = live_traffic.get('gemm4_hbm_bytes_per_launch', live_traffic.get('gemm_hbm_bytes_per_launch'))
                src = 'gpurun_out/gemm_traffic.json, ' + live_traffic['_source']
            for name in (TRAFFIC_PROFILES if traffic is None else ()):
                try:
                    with open(os.path.join(ROOT, 'profiles', name)) as f:
                        tj = json.load(f)
                    traffic = tj.get('gemm4_hbm_bytes_per_launch', tj.get('gemm_hbm_bytes_per_launch_full_depth_mix', tj.get('gemm_hbm_bytes_per_launch')))
                    src = name
                    if traffic is not None:
                        break
                except (OSError, ValueError):
                    continue          # the profile is not in this checkout: traffic stays null
            # dominant kernel = the gemm4 kernels: every launch of >= 0.25 TFLOP (all decoder / lm_head shapes of the step; they are 99.8 % of
            # the GEMM FLOPs).  The small launches (CLIP tower, projector: gemm_kernel of csrc/gemm.hip) are reported beside it -- they run at
            # the start of the step, partly beside the previous step's asynchronous AdamW on the other stream, so their event-pair times
            # include waiting for CUs and would otherwise shift the dominant kernel's figure by the luck of the sample.
            big = [(ops.event_elapsed_ms(e0, e1), fl, by) for e0, e1, fl, by in gemm_events if fl >= 0.25e12]
            if big:
                all_ach, all_n, all_ms = ach, n, tot_ms
                tot_ms, tot_fl, n = sum(b[0] for b in big), sum(b[1] for b in big), len(big)
                ach = tot_fl / tot_ms / 1e9
                gemm_events = [(None, None, fl, by) for _, fl, by in big]
            out['roofline'] = {'bound': 'mfma', 'kernel': 'gemm4_kernel<A_T,B_N,EPI> / gemm4nt_kernel<EPI> (csrc/gemm4.hip: one wave per SIMD, 128x128 per wave): '
                                                          'every sampled GEMM launch of >= 0.25 TFLOP in the timed steps',
                               'achieved': ach, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_BF16_TFLOPS,
                               'traffic': traffic, 'traffic_source': ('measured_in_this_invocation' if live_traffic is not None else
                                                                      (f'committed:profiles/{src}' if traffic is not None else None)),
                               'traffic_unit': f'HBM bytes per gemm4 launch (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, separate passes of this command: {src if live_traffic is not None else "profiles/" + str(src)})',
                               'algorithmic_bytes_per_launch': sum(e[3] for e in gemm_events) / n,
                               'launches': n, 'launch_sampling': f'every {ops.GEMM_PROF_STRIDE}th GEMM launch of the timed steps', 'avg_launch_ms': tot_ms / n,
                               'avg_flops_per_launch': tot_fl / n,
                               'gemm_share_of_step_time': tot_ms * ops.GEMM_PROF_STRIDE / (dt * 1e3),     # sample scaled to all launches
                               'by_kind_top12': by_kind}
            if power is not None:
                # measured over exactly the timed region: what "power-limited" means in numbers.  peak_at_sclk = the dense bf16 peak scaled by the mean
                # shader clock over the guide's 2400 MHz; frac_of_peak_at_sclk = the dominant kernel against THAT (how much of the clock it was
                # given it turned into MFMA work); j_per_tflop = package energy per executed TFLOP of the whole step
                out['roofline'].update(power)
                if power.get('peak_at_sclk'):
                    out['roofline']['frac_of_peak_at_sclk'] = ach / power['peak_at_sclk']
                if power.get('power_w_mean'):
                    out['roofline']['j_per_tflop_step'] = power['power_w_mean'] / exe_tflops
                    out['roofline']['joules_per_step'] = power['power_w_mean'] * step_s
            if big:
                out['roofline']['all_gemm_launches_sampled'] = {'achieved': all_ach, 'launches': all_n, 'avg_launch_ms': all_ms / all_n,
                                                                'note': 'incl. the small CLIP-tower / projector GEMMs (see the comment in bench.py)'}
                out['roofline']['gemm_share_of_step_time'] = all_ms * ops.GEMM_PROF_STRIDE / (dt * 1e3)
        if kprof and 'roofline' in out:
            # VERDICT r5 next #3.  Per kind: the sampled launches of the timed steps, ALGORITHMIC bytes of each (ops.py states them next to the launch),
            # HIP-event duration on the launch stream.  rocprofv3's per-kernel averages of the same command: profiles/r06_dpo7b_kernel_stats.csv.
            names = {'adamw_flat': 'adamw_kernel (csrc/optim.hip; 28 B per bf16 parameter: g, fp32 master / m / v read + written, bf16 weight written)',
                     'grad_sumsq': 'sumsq_kernel (csrc/optim.hip; the clip norm: one read of the gradients)',
                     'rmsnorm_fwd': 'rmsnorm_fwd2_kernel (csrc/elementwise.hip)', 'rmsnorm_bwd': 'rmsnorm_bwd_kernel<2> (csrc/elementwise.hip)',
                     'swiglu_bwd': 'swiglu_bwd_kernel (csrc/elementwise.hip)'}       # (only when the caller launches it itself: MoE / unfused paths)
            hbm = []
            for kind, label in names.items():
                ev = kprof.get(kind)
                if not ev:
                    continue
                ms = [ops.event_elapsed_ms(e0, e1) for e0, e1, _, _ in ev]
                by = sum(e[2] for e in ev)
                calls = ops._kseq.get(kind, 0)
                hbm.append({'kernel': label, 'sampled_launches': len(ev), 'algorithmic_bytes': by / len(ev), 'avg_ms': sum(ms) / len(ms),
                            'gb_per_s': by / sum(ms) / 1e6, 'frac_of_8TBs': by / sum(ms) / 1e6 / PEAK_HBM_GBS,
                            'launches_per_step': calls / args.steps, 'ms_per_step': sum(ms) / len(ms) * calls / args.steps})
            # swiglu_bwd_kernel is launched INSIDE aa_gemm_glu_bwd_bf16 when the per-shape plan chose the unfused pair: no host event can go around it.  Timed
            # here, outside the timed region, at the step's own shape with HIP events on the launch stream (its in-step average: profiles/r06_dpo7b_kernel_stats.csv)
            if any(a > b_ for _, _, _, a, b_ in ops.GLU_BWD_PROBE_LOG):
                Mtok, Fw = 2 * B * T, cfg['text']['intermediate_size']
                gu_ = torch.randn(Mtok, 2 * Fw, device=device, dtype=torch.bfloat16)
                da_ = torch.randn(Mtok, Fw, device=device, dtype=torch.bfloat16)
                o_ = torch.empty_like(gu_)
                for _ in range(3):
                    ops.swiglu_bwd(gu_, da_, out=o_)
                e0_ = ops.event_record()
                for _ in range(10):
                    ops.swiglu_bwd(gu_, da_, out=o_)
                e1_ = ops.event_record()
                ms_ = ops.event_elapsed_ms(e0_, e1_) / 10
                by_ = 2.0 * Mtok * Fw * 5
                hbm.append({'kernel': 'swiglu_bwd_kernel (csrc/elementwise.hip)', 'sampled_launches': 10, 'algorithmic_bytes': by_, 'avg_ms': ms_, 'gb_per_s': by_ / ms_ / 1e6,
                            'frac_of_8TBs': by_ / ms_ / 1e6 / PEAK_HBM_GBS, 'launches_per_step': args.layers, 'ms_per_step': ms_ * args.layers,
                            'where': 'standalone after the timed region (in the step it is launched inside aa_gemm_glu_bwd_bf16)'})
                del gu_, da_, o_
            for k_ in hbm:      # the guide's own ceiling for a streaming kernel on this part: 6.29 TB/s measured float4 copy (MI355X_MICROARCH.md, HBM3E row) against the 8 TB/s spec
                k_['frac_of_measured_copy_6290GBs'] = k_['gb_per_s'] / MEASURED_COPY_GBS
            out['roofline']['hbm_kernels'] = hbm
            att = {}
            for kind, label in (('attn_fwd', 'fwd'), ('attn_bwd', 'bwd')):
                ev = kprof.get(kind)
                if not ev:
                    continue
                ms = [ops.event_elapsed_ms(e0, e1) for e0, e1, _, _ in ev]
                fl = sum(e[3] for e in ev)
                att[label + '_tflops'] = fl / sum(ms) / 1e9
                att[label + '_avg_ms'] = sum(ms) / len(ms)
                att[label + '_frac'] = fl / sum(ms) / 1e9 / PEAK_BF16_TFLOPS
                att[label + '_sampled_launches'] = len(ev)
                att[label + '_ms_per_step'] = sum(ms) / len(ms) * ops._kseq.get(kind, 0) / args.steps
            if att:
                att['note'] = ('algorithmic FLOPs: causal 4 T^2 hd / 2 per (row, head) forward, 2.5 x that backward (five matmuls); the policy, reference-model and '
                               'launches of the step are in the forward sample (the head_dim-64 non-causal launches are a kind of their own and not reported)')
                out['roofline']['attention'] = att
        if power is not None and 'roofline' not in out:
            out['power'] = power
        if per_batch:
            per_batch[f'B{B}'] = {'pairs_per_gpu_step': B, 'value': value / world, 'unit': 'pairs/s', 'ms_per_step': step_s * 1e3, 'steps': args.steps,
                                  'warmup': args.warmup, 'headline': True}
            per_batch['note'] = (f'headline = {B} pairs/GPU/step = the largest micro-batch SURVEY.md section 8(d) allows (B in 1, 2, 4).  The reference yaml default is '
                                 'per_device_train_batch_size 1 x gradient_accumulation_steps 1 (configs/train/text_image_to_text/dpo.yaml:30,34): a user who drops the '
                                 'native trainer in with the STOCK yaml gets the B1 line below, not the headline; the headline needs per_device_train_batch_size '
                                 f'{B} in the yaml (activations for {B} pairs fit the 288 GB of one MI355X without recomputation; longer GEMM M = fewer partial tile '
                                 f'rounds, one AdamW per {B} pairs).  B1 / B2 = the same step at the yaml default micro-batch and at 2, measured after the timed region.')
            out['per_batch'] = per_batch
        if shared_prompt:
            shared_prompt['frac_of_mfma_peak_on_executed_flops'] = shared_prompt['executed_tflop_per_pair'] * shared_prompt['value'] / PEAK_BF16_TFLOPS
            out['shared_prompt'] = shared_prompt
        if ops.GLU_BWD_PROBE_LOG:
            out['glu_bwd_plan'] = [{'M': m, 'F': f, 'K': k, 'fused_ms': round(a, 4), 'unfused_ms': round(b, 4), 'chosen': 'fused' if a <= b else 'unfused'}
                                   for m, f, k, a, b in ops.GLU_BWD_PROBE_LOG]
        if multi is not None:
            out['multi_gpu'] = multi
        if not args.no_cpu_baseline and world == 1:
            try:
                out['cpu_baseline'] = cpu_baseline(cfg, T)
            except Exception as ex:  # the bench line must still be printed
                out['cpu_baseline'] = {'error': repr(ex)}
            # kind "reference": the reference's OWN DPOTrainer.loss + backward + AdamW on HF modules, timed in the build container (the
            # reference is not on the GPU box) by tools/cpu_reference_baseline.py and committed; carried beside the live port figure
            try:
                with open(os.path.join(ROOT, 'profiles', 'cpu_reference_L2.json')) as f:
                    ref = json.load(f)
                out['cpu_baseline']['reference'] = {k: ref[k] for k in ('kind', 'value', 'unit', 'cores', 'cpu', 'where', 'dtype', 'what', 'sample', 'script') if k in ref}
                out['cpu_baseline']['reference']['source'] = 'committed:profiles/cpu_reference_L2.json (not measured on this box)'
            except (OSError, ValueError):
                pass
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
