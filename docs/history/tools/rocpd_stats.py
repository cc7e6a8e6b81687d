#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace like `--stats` does: per-kernel calls, total/avg/min/max
duration, % of GPU kernel time.   python tools/rocpd_stats.py results.db [--top 40] > profiles/xxx_kernel_stats.csv"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name if len(name) <= 110 else name[:107] + '...'


def main():
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 60
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = db.execute(f'select {name_col}, (end - start), start from kernels order by start').fetchall()
    # --after-nth NAME N : keep only dispatches that start after the N-th dispatch whose name contains NAME
    # (e.g. the optimizer's last kernel of the N-th warm-up step), so that one-off MIOpen find kernels and
    # warm-up steps do not pollute the per-step statistics
    if '--after-nth' in sys.argv:
        i = sys.argv.index('--after-nth')
        key, nth = sys.argv[i + 1], int(sys.argv[i + 2])
        seen, cut = 0, None
        for n, d, st in rows:
            if key in n:
                seen += 1
                if seen == nth:
                    cut = st
                    break
        if cut is None:
            raise SystemExit(f'{key} seen {seen} times < {nth}')
        rows = [r for r in rows if r[2] > cut]
    rows = [(n, d) for n, d, _ in rows]
    agg = {}
    for n, d in rows:
        a = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print('Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs')
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f'"{n}",{a[0]},{a[1]},{a[1] / a[0]:.1f},{100.0 * a[1] / total:.3f},{a[2]},{a[3]}')
    print(f'"TOTAL ({len(agg)} distinct kernels)",{sum(a[0] for a in agg.values())},{total},,100.0,,')


if __name__ == '__main__':
    main()
