"""Worker of tests/test_dp_gpu.py::test_c_abi_collectives_two_ranks_on_one_device: rank r of 2, both on cuda:0, drives the C-ABI collectives
(csrc/comm.hip) over RCCL.  Prints ONE line: `COMM_OK ...` (all-reduce / metrics / broadcast verified) or `COMM_REFUSED <RCCL message>`."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from align_anything_amd.lib import LIB, AAHipError, call  # noqa: E402


def main():
    rank, idfile = int(sys.argv[1]), sys.argv[2]
    LIB.load()
    torch.cuda.set_device(0)
    uid = (ctypes.c_char * 128)()
    if rank == 0:
        call('aa_comm_unique_id', ctypes.cast(uid, ctypes.c_void_p))
        with open(idfile + '.tmp', 'wb') as f:
            f.write(bytes(uid))
        os.replace(idfile + '.tmp', idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 60:
                print('COMM_REFUSED rank 1 never saw the unique id', flush=True)
                return
            time.sleep(0.05)
        uid = (ctypes.c_char * 128).from_buffer_copy(open(idfile, 'rb').read())
    try:
        call('aa_comm_init', ctypes.cast(uid, ctypes.c_void_p), rank, 2)
    except AAHipError as e:
        print('COMM_REFUSED ' + str(e).replace('\n', ' '), flush=True)
        return
    st = torch.cuda.current_stream().cuda_stream
    g16 = torch.full((1 << 20,), float(rank + 1), device='cuda').to(torch.bfloat16)
    g32 = torch.arange(4097, device='cuda', dtype=torch.float32) * (rank + 1)
    m = torch.tensor([1.0, 2.0, 3.0], device='cuda') * (rank + 1)
    mx = m.clone()
    b = torch.full((1000,), float(rank), device='cuda')
    call('aa_grad_allreduce_bucket', g16.data_ptr(), g16.numel(), 0, st)
    call('aa_grad_allreduce_bucket', g32.data_ptr(), g32.numel(), 1, st)
    call('aa_metrics_allreduce', m.data_ptr(), m.numel(), 0, st)
    call('aa_metrics_allreduce', mx.data_ptr(), mx.numel(), 1, st)
    call('aa_broadcast', b.data_ptr(), b.numel() * 4, 1, st)
    torch.cuda.synchronize()
    ok = bool((g16.float() == 3.0).all()) and torch.equal(g32, torch.arange(4097, device='cuda', dtype=torch.float32) * 3) and \
        torch.allclose(m, torch.tensor([1.5, 3.0, 4.5], device='cuda')) and torch.equal(mx, torch.tensor([2.0, 4.0, 6.0], device='cuda')) and bool((b == 1.0).all())
    call('aa_comm_destroy')
    print(('COMM_OK' if ok else 'COMM_WRONG') + f' rank {rank}', flush=True)


if __name__ == '__main__':
    main()
