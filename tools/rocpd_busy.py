#!/usr/bin/env python3
"""GPU-busy analysis of a rocprofv3 (rocpd sqlite) kernel trace: wall time of the window, union of kernel intervals
(time with >= 1 kernel running), time with >= 2 kernels running, sum of kernel durations.
    python tools/rocpd_busy.py results.db [--after-nth NAME N]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = db.execute(f'select {name_col}, start, end from kernels order by start').fetchall()
    if '--after-nth' in sys.argv:
        i = sys.argv.index('--after-nth')
        key, nth = sys.argv[i + 1], int(sys.argv[i + 2])
        seen, cut = 0, None
        for n, st, en in rows:
            if key in n:
                seen += 1
                if seen == nth:
                    cut = en
                    break
        rows = [r for r in rows if cut is None or r[1] >= cut]
    ev = []
    for _, st, en in rows:
        ev.append((st, 1))
        ev.append((en, -1))
    ev.sort()
    depth, last, busy1, busy2 = 0, ev[0][0], 0, 0
    for t, d in ev:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
        depth += d
        last = t
    wall = ev[-1][0] - ev[0][0]
    tot = sum(en - st for _, st, en in rows)
    print(f'kernels {len(rows)}  window {wall / 1e6:.2f} ms  busy(>=1) {busy1 / 1e6:.2f} ms ({100 * busy1 / wall:.1f} %)  '
          f'overlapped(>=2) {busy2 / 1e6:.2f} ms ({100 * busy2 / wall:.1f} %)  sum of durations {tot / 1e6:.2f} ms')


if __name__ == '__main__':
    main()
