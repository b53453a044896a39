"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table, the same
content `rocprofv3 --stats` prints: calls, total/avg/min/max duration, share.  Usage:
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r'\(.*$', '', name)
    name = re.sub(r'^void ', '', name)
    return name[:110]


def main(db, out=None):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
    namecol = 'name' if 'name' in cols else [x for x in cols if 'name' in x][0]
    rows = c.execute(f'select {namecol}, start, end from kernels').fetchall()
    agg = {}
    for n, s, e in rows:
        k = short(n)
        d = (e - s) / 1e3  # us
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = ['kernel,calls,total_us,avg_us,min_us,max_us,percent']
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f'"{k}",{a[0]},{a[1]:.1f},{a[1]/a[0]:.2f},{a[2]:.2f},{a[3]:.2f},{100*a[1]/tot:.2f}')
    text = '\n'.join(lines) + f'\n# total kernel time {tot/1e3:.2f} ms over {len(rows)} dispatches\n'
    if out:
        open(out, 'w').write(text)
    print(text)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
