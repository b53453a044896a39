"""Worker for tests/test_dp_gpu.py: rank r of a 2-rank job runs ONE native DPO train_step on pair r of the tiny OPT
fixture (all ranks share cuda:0, gloo collectives -- the 1-GPU box has no second device for RCCL), then rank 0
saves its updated fp32 master weights.  Launched by torch.distributed.run."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_model_gpu import _batch, _trainer  # noqa: E402
from tests.util import load_golden, tiny_opt_cfg  # noqa: E402


def main():
    out = sys.argv[1]
    rank = int(os.environ['RANK'])
    torch.cuda.set_device(0)
    import datetime
    # a rank that dies must fail its peer within minutes, not after gloo's default half hour (the GPU suite runs under the driver's clock)
    dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=240))
    z = load_golden('opt_tiny_dpo.npz')
    b = _batch(z, with_pixels=False)
    rows = [rank, rank + 2]          # pair i = (chosen i, rejected i)
    mb = {'input_ids': b['input_ids'][rows], 'attention_mask': b['attention_mask'][rows],
          'meta_info': {'response_lens': [b['meta_info']['response_lens'][r] for r in rows]}}
    tr = _trainer(z, tiny_opt_cfg())
    assert tr.model.world == 2
    info = tr.train_step(mb)
    tr.model.wait_optimizer()
    torch.cuda.synchronize()
    state = {g: m.cpu() for g, m in tr.policy.store.master.items()}
    gathered = [None, None]
    dist.all_gather_object(gathered, {k: float(v.double().sum()) for k, v in state.items()})
    if rank == 0:
        assert gathered[0] == gathered[1], 'replicas diverged after one DP step'
        torch.save({'master': state, 'info': info}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
