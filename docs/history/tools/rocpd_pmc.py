#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd sqlite file.
    python tools/rocpd_pmc.py results.db [name-filter]      ->  kernel, launches, avg value (counter units)"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else 'cfl_'
    rows = db.execute('select kernel_name, counter_name, value from counters_collection').fetchall()
    agg = {}
    for name, ctr, val in rows:
        if flt not in name:
            continue
        short = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0].replace('void ', '')
        a = agg.setdefault((short, ctr), [0, 0.0])
        a[0] += 1
        a[1] += val
    print('Kernel,Counter,Launches,AvgValue')
    for (k, c), (n, s) in sorted(agg.items()):
        print(f'{k},{c},{n},{s / n:.3f}')


if __name__ == '__main__':
    main()
