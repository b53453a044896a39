"""Generator of the hand-placed K-tile schedule of csrc/gemm4.hip (one wave per SIMD, 128 x 128 per wave).

Emits csrc/gemm4_sched_{nt,nn,tn}.inc: the body of ONE K-tile = two phases of 64 MFMAs ("slots"), each slot followed by at most one
"filler" (an LDS fragment read of the NEXT fragment set, or one LDS-DMA piece of the K-tile two ahead), so that no gap between two
MFMAs carries more than one memory instruction.  Everything is inline asm placed in program order (volatile asm is not reordered);
`__builtin_amdgcn_sched_barrier(0)` after every slot keeps the compiler's own address arithmetic inside its slot.

    phase A:  MFMA(F0)  |  reads F1 <- buffer cur, k-step 1                     (all reads issued within the first 48 slots)
    s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier       (K-tile t+1 landed; everyone is done reading buffer cur)
    phase C:  MFMA(F1)  |  reads F0 <- buffer cur^1, k-step 0  |  DMA K-tile t+2 -> buffer cur   (within the first 48 slots)

LDS reads are asm, so their completion is counted here: LDS returns in order, so before the MFMAs of accumulator row i of phase A
`s_waitcnt lgkmcnt(N)` with N = (reads issued after fragment a[i] of F0) is exact (clamped to the 4-bit field: waiting for more is
safe).  Fragment order of the reads: b0..b7 then a0..a7 (row 0 needs every B fragment).

Run: python tools/gen_gemm4_sched.py   (rewrites the three .inc files; they are committed)."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'align_anything_amd', 'csrc')


def fillers(n_reads, n_dma, variant=0):
    """Proportional merge of reads and DMA pieces over the first 48 slots: slot -> filler tag ('R', k) / ('D', d).
    variant 1: the DMA pieces take slots 0..15 (the request of K-tile t+2 gets the longest possible flight time), the reads follow."""
    total = n_reads + n_dma
    assert total <= 48
    if variant == 1 and n_dma:
        slots = {d: ('D', d) for d in range(n_dma)}
        for r in range(n_reads):
            slots[n_dma + r * (48 - n_dma) // n_reads] = ('R', r)
        assert len(slots) == total
        return slots
    seq = []
    r = d = 0
    for f in range(total):          # Bresenham-style interleave: keep reads / dma in proportion
        if d * n_reads <= r * n_dma and d < n_dma and (r > 0 or n_reads == 0):
            seq.append(('D', d)); d += 1
        elif r < n_reads:
            seq.append(('R', r)); r += 1
        else:
            seq.append(('D', d)); d += 1
    slots = {}
    for f, item in enumerate(seq):
        slots[f * 48 // total] = item
    assert len(slots) == total
    return slots


def gen(a_t, b_n, variant=0):
    rA, rB = (2 if a_t else 1), (2 if b_n else 1)
    # flattened read list of one fragment set: (operand, fragment index, half)
    reads = [('b', j, h) for j in range(8) for h in range(rB)] + [('a', i, h) for i in range(8) for h in range(rA)]
    n = len(reads)
    first_read_after_a = {i: 8 * rB + (i + 1) * rA for i in range(8)}      # index of the first read issued after fragment a[i]
    L = []
    w = L.append

    def read_stmt(dst_set, k, phase_kk, buf_expr):
        op, idx, h = reads[k]
        tr = (a_t if op == 'a' else b_n)
        if not tr:
            # K-contiguous image: ds_read_b128, address = v<op>k[kk] + buffer, immediate = fragment * 2048
            return f'G4_RDK({op}{dst_set}[{idx}], v{op}k{phase_kk}_{buf_expr}, {idx * 2048});'
        # row-contiguous image: two transpose reads per fragment, address = t<op>[idx] + buffer, immediate = kk*16384 + half*2048
        off = 'cbc' if buf_expr == 'cur' else 'cbn'
        return f'G4_RDT({op}{dst_set}h[{idx}][{h}], t{op}[{idx}] + {off}, {phase_kk * 16384 + h * 2048});'

    def mfma(setn, i, j):
        af = f'G4_FRAG_A({setn}, {i})'
        bf = f'G4_FRAG_B({setn}, {j})'
        return f'G4_MFMA(acc[{i}][{j}], {bf}, {af});'

    # ---------------- phase A: MFMA(F0), reads F1 from buffer `cur` (k-step 1); waits for the F0 reads issued in the last phase C
    fa = fillers(n, 0)
    issued_new = 0
    for s in range(64):
        i, j = s // 8, s % 8
        if j == 0:
            # F0 fragments a[i] (and every b for i == 0) must have landed: reads issued after a[i] in the previous phase C + the new
            # reads issued so far in this phase may still be outstanding
            pend = (n - first_read_after_a[i]) + issued_new
            if variant not in (3, 4):
                w(f'G4_WAIT_LGKM({min(pend, 15)});')
        w(mfma(0, i, j))
        if s in fa:
            if variant not in (3, 4):
                w(read_stmt(1, fa[s][1], 1, 'cur'))
            issued_new += 1
        w('G4_PIN;')
    assert issued_new == n
    w('G4_SYNC;')
    # ---------------- phase C: MFMA(F1) (complete: waited at the barrier), reads F0 from buffer cur^1 (k-step 0), DMA into buffer cur
    fc = fillers(n, 16, variant)
    for s in range(64):
        i, j = s // 8, s % 8
        w(mfma(1, i, j))
        if s in fc:
            kind, k = fc[s]
            if kind == 'R' and variant not in (3, 4):
                w(read_stmt(0, k, 0, 'nxt'))
            if kind == 'D' and variant not in (2, 4):
                w(f'G4_DMA({k});')
        w('G4_PIN;')
    name = ('t' if a_t else 'n') + ('n' if b_n else 't')
    path = os.path.join(OUT, f'gemm4_sched_{name}' + (f'_v{variant}' if variant else '') + '.inc')
    hdr = (f'// GENERATED by tools/gen_gemm4_sched.py -- do not edit.  K-tile schedule of gemm4_kernel<{str(a_t).lower()}, {str(b_n).lower()}>:\n'
           f'// {n} LDS reads per fragment set ({rA} per A fragment, {rB} per B fragment), 16 DMA pieces per K-tile, at most one filler per MFMA slot.\n')
    with open(path, 'w') as f:
        f.write(hdr + '\n'.join(L) + '\n')
    return path, n


if __name__ == '__main__':
    import sys
    # variant 0 is the build; `python tools/gen_gemm4_sched.py 1` (DMA pieces first: measured 7-10 % slower) and 2 / 3 / 4 (ablations
    # without DMA / without LDS reads / without both: timing only) write gemm4_sched_*_v<N>.inc for experiments
    for variant in [int(a) for a in sys.argv[1:]] or [0]:
        for a_t, b_n in ((False, False), (False, True), (True, True)):
            print(*gen(a_t, b_n, variant))
