"""GPU: pin the two hardware layout facts every MFMA kernel here is built on."""
import pytest
import torch

from tests.gpu_util import dev, dump

pytestmark = pytest.mark.gpu


def test_mfma_16x16x32_output_layout():
    """D[row][col]: lane holds col = lane & 15, rows (lane >> 4) * 4 + r (cdna guide §3)."""
    from align_anything_amd.lib import call
    lines = []
    for row_sel in (0, 5, 10, 15):
        out = torch.zeros(256, device=dev())
        call('aa_probe_mfma', out.data_ptr(), row_sel, torch.cuda.current_stream().cuda_stream)
        o = out.cpu().view(64, 4)
        lines.append(f'row_sel={row_sel}\n{o}')
        for lane in range(64):
            for r in range(4):
                want = float((lane & 15) + 1) if (lane >> 4) * 4 + r == row_sel else 0.0
                assert o[lane, r].item() == want, (row_sel, lane, r, o[lane, r].item(), want)
    dump('probe_mfma.txt', '\n'.join(lines))


def test_ds_read_b64_tr_b16_lane_mapping():
    """Within each 16-lane group: result[lane i][j] = (8 bytes loaded by lane 4j + i/4)[i % 4]."""
    from align_anything_amd.lib import call
    st = torch.cuda.current_stream().cuda_stream
    # pattern A: contiguous 4x16 blocks (row = 32 B): lane l -> byte (l>>4)*128 + (l&15)*8
    addr = torch.tensor([(l >> 4) * 128 + (l & 15) * 8 for l in range(64)], dtype=torch.int32, device=dev())
    out = torch.zeros(256, device=dev())
    call('aa_probe_tr16', addr.data_ptr(), out.data_ptr(), st)
    a = out.cpu().view(64, 4)
    # pattern B: rows 64 B apart: lane i -> row (i>>2), col (i&3)*4 ; group g offset 1024 B
    addr_b = torch.tensor([(l >> 4) * 1024 + ((l & 15) >> 2) * 64 + (l & 3) * 8 for l in range(64)],
                          dtype=torch.int32, device=dev())
    out_b = torch.zeros(256, device=dev())
    call('aa_probe_tr16', addr_b.data_ptr(), out_b.data_ptr(), st)
    b = out_b.cpu().view(64, 4)
    dump('probe_tr16.txt', f'A (contiguous blocks):\n{a}\nB (strided rows):\n{b}\n')
    for l in range(64):
        for j in range(4):
            elem = (l >> 4) * 64 + j * 16 + (l & 15)
            assert a[l, j].item() == float(elem & 255), ('A', l, j, a[l, j].item(), elem & 255)
            elem_b = ((l >> 4) * 1024 + j * 64) // 2 + (l & 15)
            assert b[l, j].item() == float(elem_b & 255), ('B', l, j, b[l, j].item(), elem_b & 255)
