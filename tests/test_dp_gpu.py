"""GPU: data-parallel step equivalence on hardware.  Two ranks (one pair each; both on cuda:0 with gloo collectives,
since the test box has one GPU) must produce the same update as ONE rank stepping on the 2-pair batch: the DP mean of
per-rank gradients == the batch-mean gradient (DeepSpeed engine semantics, trainers/text_to_text/dpo.py:205-237),
and the logged loss is the all-reduce mean (utils/multi_process.py:74-89)."""
import os
import subprocess
import sys

import pytest
import torch

from tests.test_model_gpu import _batch, _trainer
from tests.util import ROOT, load_golden, tiny_opt_cfg

pytestmark = pytest.mark.gpu


def test_two_rank_dp_step_equals_single_rank_full_batch(tmp_path):
    out = str(tmp_path / 'dp2.pt')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(ROOT, 'tests', 'dp_worker.py'), out]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        from tests.gpu_util import dump
        dump('dp_worker_failure.log', r.stdout + '\n' + r.stderr)
    assert r.returncode == 0, r.stderr[-3000:]
    dp = torch.load(out)
    z = load_golden('opt_tiny_dpo.npz')
    full = _trainer(z, tiny_opt_cfg())
    info = full.train_step(_batch(z, with_pixels=False))
    full.model.wait_optimizer()
    torch.cuda.synchronize()
    assert abs(dp['info']['train/loss'] - info['train/loss']) < 2e-3, (dp['info']['train/loss'], info['train/loss'])
    for g, f in full.policy.store.master.items():
        a, f = dp['master'][g], f.cpu()
        # Adam's first step moves every weight by ~lr = 1e-3; a sign flip of a near-zero gradient component costs 2e-3
        assert (a - f).abs().max().item() <= 2.1e-3, (g, (a - f).abs().max().item())
        assert ((a - f).abs() < 1e-4).float().mean().item() > 0.97, g


def test_c_abi_collectives_bind_rccl_and_run_on_one_rank():
    """csrc/comm.hip: aa_comm_unique_id / aa_comm_init / aa_grad_allreduce_bucket / aa_metrics_allreduce / aa_broadcast bind librccl at
    run time and work as a 1-rank communicator on the test box's one GPU (the N-rank behaviour is RCCL's; what is pinned here is that
    the symbols resolve, the communicator comes up on the current device and the calls are stream-ordered and leave a 1-rank job's data
    untouched)."""
    import ctypes
    from align_anything_amd.lib import LIB, AAHipError, call
    dll = LIB.load()
    uid = (ctypes.c_char * 128)()
    call('aa_comm_unique_id', ctypes.cast(uid, ctypes.c_void_p))
    assert any(b != b'\x00' for b in uid)
    torch.cuda.set_device(0)
    call('aa_comm_init', ctypes.cast(uid, ctypes.c_void_p), 0, 1)
    try:
        with pytest.raises(AAHipError):
            call('aa_comm_init', ctypes.cast(uid, ctypes.c_void_p), 0, 1)       # one communicator per process
        r, w = ctypes.c_int(-1), ctypes.c_int(-1)
        call('aa_comm_world', ctypes.byref(r), ctypes.byref(w))
        assert (r.value, w.value) == (0, 1)
        st = torch.cuda.current_stream().cuda_stream
        g16 = torch.randn(1 << 20, device='cuda').to(torch.bfloat16)
        g32 = torch.randn(4097, device='cuda')
        m = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0], device='cuda')
        keep = (g16.clone(), g32.clone(), m.clone())
        call('aa_grad_allreduce_bucket', g16.data_ptr(), g16.numel(), 0, st)
        call('aa_grad_allreduce_bucket', g32.data_ptr(), g32.numel(), 1, st)
        call('aa_metrics_allreduce', m.data_ptr(), m.numel(), 0, st)
        call('aa_metrics_allreduce', m.data_ptr(), m.numel(), 1, st)
        call('aa_broadcast', g32.data_ptr(), g32.numel() * 4, 0, st)
        torch.cuda.synchronize()
        assert torch.equal(g16, keep[0]) and torch.equal(g32, keep[1]) and torch.equal(m, keep[2])
        with pytest.raises(AAHipError):
            call('aa_grad_allreduce_bucket', g16.data_ptr(), g16.numel(), 7, st)
    finally:
        call('aa_comm_destroy')


def test_c_abi_collectives_two_ranks_on_one_device(tmp_path):
    """drive aa_comm_init / aa_grad_allreduce_bucket / aa_metrics_allreduce / aa_broadcast with WORLD 2 on the test box's
    one GPU (two processes, both on cuda:0).  RCCL either brings the communicator up -- then sums, means, maxima and the broadcast are checked
    on both ranks -- or refuses two ranks on one device (NCCL's "duplicate GPU" rule); which of the two happened on this box is written to
    gpurun_out/comm_two_ranks_one_device.txt.  A wrong value fails the test; both workers run under a timeout (a bootstrap that blocks on
    the duplicate device is recorded and skipped)."""
    from tests.gpu_util import dump
    idfile = str(tmp_path / 'rccl_uid.bin')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', NCCL_DEBUG='WARN')
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'comm_worker.py'), str(r), idfile], env=env, cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p_ in procs:
        try:
            o, e = p_.communicate(timeout=90)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            o, e = p_.communicate()
            o += '\nCOMM_TIMEOUT'
        outs.append((p_.returncode, o, e))
    lines = []
    for rc, o, e in outs:
        import re
        mine = re.findall(r'COMM_(?:OK|REFUSED|WRONG|TIMEOUT)[^\n]*', o)      # RCCL's own warnings share the stream: the marker may sit mid-line
        # a worker that dies inside librccl without reaching a print (RCCL aborting on the duplicate device) counts as a refusal, with its exit code
        lines.append(mine[0] if mine else f'COMM_REFUSED process ended with rc {rc} before reporting: {(e.strip().splitlines() or ["(no stderr)"])[-1][:300]}')
    dump('comm_two_ranks_one_device.txt', '\n'.join(lines) + '\n\n' + '\n'.join(f'--- rank {i} rc {rc}\n{o[-1500:]}\n{e[-3000:]}' for i, (rc, o, e) in enumerate(outs)))
    assert not any(ln.startswith('COMM_WRONG') for ln in lines), lines
    if any(ln.startswith('COMM_TIMEOUT') for ln in lines):
        # two ranks on ONE device is outside what RCCL supports: a bootstrap that blocks instead of refusing is this box's RCCL, not csrc/comm.hip
        # (whose N-rank use is one rank per GPU); recorded in the evidence file above, not a verdict on the wrapper
        pytest.skip('RCCL neither came up nor refused two ranks on one device within the timeout: ' + ' | '.join(lines))
    ok = [ln.startswith('COMM_OK') for ln in lines]
    assert all(ok) or not any(ok), lines          # both ranks up and correct, or RCCL refused the duplicate device on both
