"""Per-kernel split of the decode window of a rocprofv3 kernel trace of tools/bench_ppo.py: the LAST contiguous run of launches between the
first and the last skinny GEMM of the timed iteration's `generate` (positions = number of sampler launches in it), busy time against the
window's span (the difference is launch gaps), per kernel: launches per position, average duration, share of the window."""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
short = lambda n: n.split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')[-70:]
# decode windows: maximal runs in which consecutive sampler launches are < 20 ms apart; keep the last one (the timed iteration)
samp = [i for i, r in enumerate(rows) if 'sample_top_p_kernel' in r[2] or 'argmax_rows_kernel' in r[2]]
runs, cur = [], [samp[0]]
for a, b in zip(samp, samp[1:]):
    if rows[b][0] - rows[a][0] < 20e6:
        cur.append(b)
    else:
        runs.append(cur); cur = [b]
runs.append(cur)
win = max(runs[-2:], key=len) if len(runs) > 1 else runs[-1]
lo, hi = win[1], win[-1]             # from the 2nd sampler launch (prefill excluded) to the last: len(win) - 2 full positions
pos = len(win) - 2
sel = rows[lo + 1:hi + 1]
span = (sel[-1][1] - sel[0][0]) / 1e3
busy = sum(e - s for s, e, _ in sel) / 1e3
gaps = sum(max(0, b[0] - a[1]) for a, b in zip(sel, sel[1:])) / 1e3
print(f'decode window: {pos} positions, {len(sel)} launches ({len(sel) / pos:.1f} per position), span {span / pos:.1f} us per position, '
      f'kernel busy {busy / pos:.1f} us per position, idle between launches {gaps / pos:.1f} us per position')
d = collections.defaultdict(list)
for s, e, n in sel:
    d[short(n)].append((e - s) / 1e3)
print(f"{'kernel':72s} {'per pos':>8s} {'avg us':>9s} {'us / pos':>9s} {'share':>7s}")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f'{k:72s} {len(v) / pos:8.1f} {sum(v) / len(v):9.2f} {sum(v) / pos:9.1f} {100 * sum(v) / span:6.1f}%')
