"""Generator of the hand-placed step schedule of csrc/attn128.inc (head_dim-128 attention forward, one wave per SIMD, 32 x 32 x 16 MFMA).

Emits csrc/attn128_sched.inc: the 32 MFMA gaps of ONE pipeline step `j` (half tile H_j = 32 keys, both 32-query blocks of the wave) for PAR = j & 1,
as straight-line statements inside `fwd_step`.  Per gap: the MFMA, then its fillers -- LDS fragment reads (compiler builtins: the compiler owns their
lgkmcnt), ONE asm statement with the gap's VALU instructions (softmax of H_j, row maxima of H_(j+1)), at most one LDS-DMA piece -- and a sched_barrier.

What the schedule is built on (tools/lab/ubench/mfma_fill.hip, cycles per v_mfma_f32_32x32x16_bf16 with F fillers per gap, one wave per SIMD):
    bare 34-35 | v_fma_f32: F = 5 -> 36.8, 6 -> 42, 8 -> 52 (about 5 cycles each, five hide) | v_exp_f32: F = 2 -> 36.6, 4 -> 56, 8 -> 83 (about 12 each)
    | exp/add/cvt mix F = 4 -> 44.5 | ds_read_b128 about 3 | two dependent accumulator chains in VGPRs: no penalty.
So a step of 32 MFMAs with 32 exponentials and ~150 other issues is close to VALU-bound (~1340 cycles against 1100 for the MFMAs): the scheduler below
spreads the stream (per score v_fma_f32 -- scale, minus running maximum --, v_exp_f32, v_add_f32 and half a v_cvt_pk_bf16_f32) under a per-gap COST cap with
at most one exponential per gap.  (Row sums as 4 extra MFMAs with an all-ones A fragment were tried: slower, the step is bound by the tile feed, below.)

The tile feed: a workgroup pulls 32 KB (one K and one V tile) per 64 MFMAs through LDS-DMA; in-kernel clocks (profiles/r04_attn128_trace.txt) showed a step
that carries 8 pieces per wave at 2255 (L2-hot GQA) .. 3063 (causal MHA) cycles against 1634 for one without, plus ~300 for a full vmcnt(0) drain per
tile.  So the pieces are a CONTINUOUS stream -- four per step, K(i + 3) in odd steps and V(i + 2) in even ones -- and the tile boundary waits with
vmcnt(4): only for the tiles the next steps read, never for the pieces just issued.

MFMA order: the 16 score MFMAs S(H_(j+1))[qb] += K_t Q_t in gaps 3t, 3t+1; O[d][qb] += V^T P^T(H_(j-1)) in gaps 3t+2 and 24..31.
Fragments are read a whole step ahead: K fragment ks (for the NEXT step's chains) in the gap after its last use, V^T fragment n likewise; V^T fragment 7
(used by the last PV MFMAs) is read in gap 0 of the step that uses it.

Run: python tools/gen_attn128_sched.py   (rewrites csrc/attn128_sched.inc; committed)."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'align_anything_amd', 'csrc', 'attn128_sched.inc')
NGAP = 32
COST = {'exp': 12, 'plain': 5, 'tr': 6, 'b128': 3, 'addr': 5, 'dma': 20}
MAX_FIRST_GAP = 25  # first gap of the row-maximum stream (the last score MFMA is gap 22)


def mfma_order():
    """gap -> ('Q', ks, qb) | ('P', i)"""
    g = {}
    for t in range(8):
        g[3 * t] = ('Q', t, 0)
        g[3 * t + 1] = ('Q', t, 1)
        g[3 * t + 2] = ('P', t)
    for i in range(8, 16):
        g[24 + i - 8] = ('P', i)
    assert len(g) == NGAP
    return g


def e_stream(par):
    """The E instructions of H_j in program order: (kind, text, writes, reads, post statement); {w0} / {r0}.. are the operands."""
    S = lambda t: f'c.s[{par}][{t >> 4}][{t & 15}]'
    ins = []
    for k in range(36):
        tf, te, ta = k, k - 1, k - 3
        if 0 <= tf < 32:
            ins.append(('plain', 'v_fma_f32 {w0}, {r0}, {r1}, {r2}', [f'fa[{tf & 1}]'], [S(tf), 'c2v', f'nms[{tf >> 4}]'], None))
        if 0 <= te < 32:
            ins.append(('exp', 'v_exp_f32 {w0}, {r0}', [f'ex[{te & 3}]'], [f'fa[{te & 1}]'], None))
        if 0 <= ta < 32:
            ins.append(('plain', 'v_add_f32 {w0}, {w0}, {r1}', [f'ps[{ta >> 4}]'], [f'ps[{ta >> 4}]', f'ex[{ta & 3}]'], None))
            if ta & 1:
                qb, r = ta >> 4, ta & 15
                ins.append(('plain', 'v_cvt_pk_bf16_f32 {w0}, {r0}, {r1}', [f'pw{ta}'], [f'ex[{(ta - 1) & 3}]', f'ex[{ta & 3}]'],
                            f'A128_SETW(c.pf[{par}][{qb}][{r >> 3}], {(r >> 1) & 3}, pw{ta});'))
    return ins


def max_stream(par):
    npar = par ^ 1
    S = lambda qb, r: f'c.s[{npar}][{qb}][{r}]'
    ins = []
    for qb in range(2):
        ins.append(('plain', 'v_max3_f32 {w0}, {r0}, {r1}, {r2}', [f'mxa[{qb}]'], [S(qb, 0), S(qb, 1), S(qb, 2)], None))
    for r in range(3, 15, 2):
        for qb in range(2):
            ins.append(('plain', 'v_max3_f32 {w0}, {w0}, {r1}, {r2}', [f'mxa[{qb}]'], [f'mxa[{qb}]', S(qb, r), S(qb, r + 1)], None))
    for qb in range(2):
        ins.append(('plain', 'v_max_f32 {w0}, {w0}, {r1}', [f'mxa[{qb}]'], [f'mxa[{qb}]', S(qb, 15)], None))
    return ins


def emit_asm(group, mfma=None):
    """ONE asm volatile statement for a list of instructions (and, first, the gap's MFMA: hipcc puts an s_nop between two adjacent asm statements whose
    registers it cannot relate, one issue slot per gap).  Every written lvalue is one output operand ("+v" when the statement reads it at or before its
    first write, "=&v" otherwise -- early clobber: later instructions of the statement still read the inputs); every other lvalue an input.
    mfma = (text with {acc} {a} {b}, acc lvalue, acc constraint letter, a lvalue, b lvalue, b constraint letter, accumulate?)"""
    written, first_write, first_read = [], {}, {}
    for pos, (kind, text, wr, rd, post) in enumerate(group):
        for l in rd:
            first_read.setdefault(l, pos)
        for l in wr:
            first_write.setdefault(l, pos)
            if l not in written:
                written.append(l)
    inputs = []
    for kind, text, wr, rd, post in group:
        for l in rd:
            if l not in written and l not in inputs:
                inputs.append(l)
    outs = [('"+v"' if l in first_read and first_read[l] <= first_write[l] else '"=&v"') + f'({l})' for l in written]
    names = list(written)
    if mfma:
        text, acc, accc, a, b, bc, accum = mfma
        outs.append((f'"+{accc}"' if accum else f'"=&{accc}"') + f'({acc})')
        names.append(acc)
    ins = [f'"v"({l})' for l in inputs]
    names += inputs
    if mfma:
        for l, cst in ((a, 'v'), (b, bc)):
            assert l not in names
            ins.append(f'"{cst}"({l})')
            names.append(l)
    opno = {l: n for n, l in enumerate(names)}
    lines = []
    if mfma:
        lines.append(text.format(acc=f'%{opno[acc]}', a=f'%{opno[a]}', b=f'%{opno[b]}'))
    for kind, text, wr, rd, post in group:
        m = {f'w{n}': f'%{opno[l]}' for n, l in enumerate(wr)}
        m.update({f'r{n}': f'%{opno[l]}' for n, l in enumerate(rd)})
        lines.append(text.format(**m))
    decls = [f'int {l};' for l in written if l.startswith('pw')]
    body = '\\n\\t'.join(lines)
    stmt = ' '.join(decls) + (' ' if decls else '') + f'asm volatile("{body}" : {", ".join(outs)} : {", ".join(ins)});'
    return [stmt] + [post for kind, text, wr, rd, post in group if post]


def place(fixed, E, M):
    """Per-gap instruction lists under the smallest cost cap that fits: M (from MAX_FIRST_GAP, in order) before E (in order), at most one exp per gap."""
    for cap, max_exp in [(c, 1) for c in range(20, 40)] + [(c, 2) for c in range(28, 80)]:
        ei = mi = 0
        out = []
        for g in range(NGAP):
            cost, grp, nexp = fixed[g], [], 0
            if g >= MAX_FIRST_GAP:
                left = NGAP - 1 - g          # gaps after this one (the last gap is reserved for the lane-half combine)
                while mi < len(M) and (cost + COST['plain'] <= cap or len(M) - mi > 2 * max(left - 1, 0)):
                    grp.append(M[mi]); mi += 1; cost += COST['plain']
            while ei < len(E):
                k = E[ei][0]
                if k == 'exp' and nexp >= max_exp:
                    break
                if cost + COST[k] > cap:
                    break
                grp.append(E[ei]); ei += 1; cost += COST[k]; nexp += k == 'exp'
            out.append((grp, cost))
        if ei == len(E) and mi == len(M):
            return cap, out
    raise RuntimeError('no placement')


def gen_body(par):
    npar = par ^ 1
    order = mfma_order()
    L = []
    w = L.append
    E = e_stream(par)
    M = max_stream(par)
    lds = {g: [] for g in range(NGAP)}
    pre = {g: [] for g in range(NGAP)}          # address arithmetic (PAR 0) that must precede the gap's reads
    dma = {}
    koff_next = par * 8192
    for ks in range(8):
        g = 3 * ks + 2
        if par == 0:
            pre[g].append(f'c.kA[{ks}] += kdelta;')
        lds[g].append(f'c.kf[{ks}] = rd128(c.kA[{ks}] + {koff_next});')
    last_use = {}
    for g, m in order.items():
        if m[0] == 'P':
            last_use[m[1] >> 1] = max(last_use.get(m[1] >> 1, 0), g)
    voff_next = par * 8192              # H_j for the next step's PV
    voff_this = npar * 8192             # H_(j-1), fragment 7 only
    def vread(n, off):
        kq, d = n >> 2, n & 3
        return f'c.vf[{n}] = rdtr2(c.tA[{d}][0] + {off + kq * 16 * 256}, c.tA[{d}][1] + {off + kq * 16 * 256});'
    lds[0].append(vread(7, voff_this))
    if par == 0:
        for d in range(4):              # the fragments of H_j live in the next V tile: step the addresses after gap 0's read
            pre[1 + d].append(f'c.tA[{d}][0] += vdelta; c.tA[{d}][1] += vdelta;')
    for n in range(7):
        g = last_use[n] + 1
        if par == 0:
            g = max(g, 6)
        assert g < NGAP
        lds[g].append(vread(n, voff_next))
    for p, g in enumerate((3, 10, 16, 22)):          # four pieces per step: K(i + 3) in odd steps, V(i + 2) in even ones -- a continuous stream
        dma[g] = p
    fixed = {}
    for g in range(NGAP):
        c = sum(COST['tr'] if 'rdtr2' in s else COST['b128'] for s in lds[g])
        c += COST['addr'] * sum(s.count('+=') for s in pre[g])
        if g in dma:
            c += COST['dma']
        if g == NGAP - 1:
            c += 3 * COST['plain']          # swap, max, nop
        fixed[g] = c
    cap, placed = place(fixed, E, M)
    for g in range(NGAP):
        m = order[g]
        if m[0] == 'Q':
            _, ks, qb = m
            w(f'A128_MFMA_S{"0" if ks == 0 else ""}(c.s[{npar}][{qb}], c.kf[{ks}], c.q[{qb}][{ks}]);')
        else:
            i = m[1]
            n, qb = i >> 1, i & 1
            w(f'A128_MFMA_O(c.o[{n & 3}][{qb}], c.vf[{n}], c.pf[{npar}][{qb}][{n >> 2}]);')
        # (the MFMA as the first line of the gap's VALU statement removes hipcc's s_nop between adjacent asm statements -- and measured 2.5 % SLOWER)
        for st in pre[g]:
            w(st)
        for st in lds[g]:
            w(st)
        if g == MAX_FIRST_GAP:
            w(f'if (mask_next) mask_half<true>(c.s[{npar}], kvh_next, qw, start, KT, causal, c);      // wave-uniform, rare: diagonal / padding boundary')
        grp, cost = placed[g]
        if grp:
            for st in emit_asm(grp):
                w(st)
        if g in dma:
            p = dma[g]
            if par == 1:
                w(f'A128_DMA_PIECE(dm.kdst, {p * 4096}, c.koff[{p}], dm.kptr);')
            else:
                w(f'A128_DMA_PIECE(dm.vdst, {p * 4096}, c.voff[{p}], dm.vptr);')
        if g == NGAP - 1:
            w('A128_ROWMAX_COMBINE(c.mxm, mxa[0], mxa[1]);')
        w(f'A128_FENCE;      // gap {g}: cost {cost} (cap {cap}), {sum(1 for x in grp if x[0] == "exp")} exp + {sum(1 for x in grp if x[0] != "exp")} plain')
    return L, cap


def main():
    out = ['// GENERATED by tools/gen_attn128_sched.py -- do not edit.  The 32 gaps of one pipeline step, included inside a128::fwd_step.']
    for par in (0, 1):
        body, cap = gen_body(par)
        out.append(f'{"if" if par == 0 else "else if"} constexpr (PAR == {par}) {{      // filler cost cap per gap: {cap}')
        out += ['    ' + s for s in body]
        out.append('}')
        print('PAR', par, 'cap', cap)
    with open(OUT, 'w') as f:
        f.write('\n'.join(out) + '\n')
    print('wrote', OUT, len(out), 'lines')


if __name__ == '__main__':
    main()
