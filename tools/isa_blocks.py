"""Instruction mix per basic block of one kernel in a hipcc -S listing: python tools/isa_blocks.py attn.s <kernel-substring> [min_mfma]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
start = [i for i, l in enumerate(lines) if re.match(r'^[_A-Za-z0-9]+:', l) and key in l][0]
end = next(i for i in range(start, len(lines)) if '.Lfunc_end' in lines[i])
bb, stats = 'entry', collections.OrderedDict()
stats[bb] = collections.Counter()
for l in lines[start:end]:
    s = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', s) or re.match(r'^; %bb\.(\d+):', s)
    if m:
        bb = m.group(0).rstrip(':')
        stats[bb] = collections.Counter()
        continue
    if not s or s.startswith(';') or s.startswith('.'):
        continue
    op = s.split()[0]
    k = ('mfma' if op.startswith('v_mfma') else 'accvgpr' if op.startswith('v_accvgpr') else 'valu' if op.startswith('v_') else 's_nop' if op.startswith('s_nop') else
         'waitcnt' if op.startswith('s_waitcnt') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'other')
    stats[bb][k] += 1
    stats[bb]['op:' + op] += 1
for b, c in stats.items():
    if c['mfma'] >= min_mfma:
        tot = sum(v for k, v in c.items() if not k.startswith('op:'))
        print(b, {k: v for k, v in c.items() if not k.startswith('op:')}, 'issues/mfma %.2f' % ((tot - c['mfma']) / c['mfma']))
        print('   ', sorted([(k[3:], v) for k, v in c.items() if k.startswith('op:')], key=lambda x: -x[1])[:24])
