"""bench.py launch contract: the driver runs `python bench.py --gpus N ...` WITHOUT a launcher, so for N > 1 the script
must spawn its own N ranks (torch.distributed.run, one per GPU); under a launcher it must use the ranks it was given.
CPU: the self-launch command.  GPU: a functional 2-rank run on the one device of the test box (gloo collectives)."""
import json
import os
import subprocess
import sys

import pytest

from tests.util import ROOT


def _run(args, env_extra, timeout):
    env = dict(os.environ, **env_extra)
    env.pop('WORLD_SIZE', None), env.pop('RANK', None), env.pop('LOCAL_RANK', None)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, cwd=ROOT, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_self_launches_n_ranks_when_no_launcher_is_present():
    r = _run(['--gpus', '4', '--steps', '3', '--warmup', '1'], {'AA_BENCH_DRYRUN_LAUNCH': '1'}, 120)
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = json.loads(r.stdout.strip().splitlines()[-1])['self_launch']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=4' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '4', '--steps', '3', '--warmup', '1']      # the user's flags reach every rank verbatim


@pytest.mark.gpu
def test_bench_two_ranks_functional_on_one_device():
    """`python bench.py --gpus 2` end to end (self-launch -> 2 ranks -> gradient buckets all-reduced -> one JSON line),
    reduced depth / length so it takes seconds; both ranks share cuda:0 and talk over gloo: a functional check only."""
    r = _run(['--gpus', '2', '--steps', '2', '--warmup', '1', '--layers', '1', '--seq-len', '1024', '--response-len', '128',
              '--pairs-per-gpu', '1', '--no-cpu-baseline', '--comm-prof'], {'AA_BENCH_ONE_DEVICE': '1', 'AA_BENCH_BACKEND': 'gloo'}, 900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(line) == 1, r.stdout[-2000:]
    out = json.loads(line[0])
    assert out['n_gpus'] == 2 and out['config']['global_batch_pairs'] == 2 and out['config']['parallelism'] == 'dp2'
    assert out['value'] > 0 and all(abs(x - 0.6931) < 0.05 for x in out['config']['losses_timed_steps'])
    assert 'rank 0/2' in r.stderr and 'rank 1/2' in r.stderr and 'world_seen_by_collective=2' in r.stderr
    # the replicas saw different batches, exchanged gradient buckets and must still hold bit-identical weights
    mg = out['multi_gpu']
    assert mg['world'] == 2 and mg['replicas_bit_identical_after_steps'] is True and mg['optimizer_updates_checked'] == 3
    assert mg['comm']['buckets'] >= 2 and mg['comm']['bytes_total'] > 0 and mg['comm']['backward_ms'] > 0
