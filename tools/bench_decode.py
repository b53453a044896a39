"""Decode (rollout) throughput of the native generate() on the LLaVA-1.5-7B text geometry (Vicuna-7B):
ms/token and the HBM rate implied by the weight stream (13.5 GB of bf16 weights per generated position)."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align_anything_amd import configs
from align_anything_amd.generation import generate
from align_anything_amd.modeling import build_model
dev = torch.device('cuda:0')
cfg = configs.llava_1_5_7b()['text']
m = build_model(cfg, dev, trainable=False)
g = torch.Generator(device=dev).manual_seed(0)
for name, s in m.store.specs.items():
    p = m.store.p[name]
    p.normal_(0, 0.02, generator=g) if len(s['shape']) >= 2 else p.fill_(1.0)
wbytes = 2.0 * sum(s['numel'] for n, s in m.store.specs.items() if 'embed_tokens' not in n)
res = []
import itertools
from align_anything_amd import ops
quick = os.environ.get('AA_BENCH_DECODE_QUICK') == '1'      # one configuration, default kernels: the run rocprofv3 wraps
cases = itertools.product(((4, 512, 32),) if os.environ.get("AA_BENCH_DECODE_AB") != "1" else ((4, 512, 64), (16, 512, 64)), (False,)) if quick else itertools.product(((4, 512, 64), (16, 512, 64), (16, 1536, 64)), (False,))
if os.environ.get('AA_BENCH_DECODE_CASES'):      # 'N,prompt,new;N,prompt,new;...' (round 6: the batches the reference rolls out -- PPO 1, GRPO B x num_generations = 10)
    cases = [(tuple(int(x) for x in c.split(',')), False) for c in os.environ['AA_BENCH_DECODE_CASES'].split(';')]
for (N, Tp, new), fused in cases:
    ops.DECODE_FUSED = fused; ug = os.environ.get('AA_BENCH_DECODE_GRAPH') == '1'
    ids = torch.randint(3, 32000, (N, Tp), device=dev)
    mask = torch.ones_like(ids)
    # every timed call is preceded by an identical untimed one: buffers of the same sizes come back from torch's caching
    # allocator instead of hipMalloc / hipFree of tens of GB inside the timed region (as in a PPO loop, where every rollout has the same shape)
    kw = dict(do_sample=True, temperature=1.0, top_p=0.9, pad_token_id=0, use_graph=ug)
    def timed(n_new):
        generate(m, ids, mask, max_new_tokens=n_new, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        generate(m, ids, mask, max_new_tokens=n_new, **kw)
        torch.cuda.synchronize(); return time.perf_counter() - t0
    t_prefill = timed(1)
    dt = timed(new + 1) - t_prefill
    ms_tok = dt / new * 1e3
    row = dict(N=N, prompt=Tp, new=new, fused=fused, strip_major_weights=os.environ.get('AA_DECODE_SWIZZLE', '1') != '0', hipgraph=ug, graph_used=generate.last_used_graph, prefill_ms=t_prefill * 1e3, ms_per_step=ms_tok, tokens_per_s=N * new / dt,
               weight_stream_GBs=wbytes / (ms_tok * 1e-3) / 1e9, frac_hbm_peak=wbytes / (ms_tok * 1e-3) / 8e12)
    print(row, flush=True); res.append(row)
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', os.environ.get('AA_BENCH_DECODE_OUT') or ('bench_decode_quick.json' if quick else 'bench_decode.json')), 'w'), indent=1)
