#!/usr/bin/env python3
"""Per-kernel HBM traffic from the two PMC passes of tools/pmc_run.sh (rocprofv3 csv output).
FETCH_SIZE / WRITE_SIZE are in KB; per MI355X_MICROARCH.md the gfx950 FETCH_SIZE tallies the 128-byte requests of wide
coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is taken as reported."""
import csv, glob, json, os, re, sys
# optional second argument: the number of steps (warm-up + timed) the profiled command ran -> `launches_per_step` per kernel,
# which bench.py compares with its own run before quoting a traffic figure (a stale file must not be paired with new code)
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 0
out = {}
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(os.path.join(sys.argv[1], ctr, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') != ctr:
                continue
            name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace('void ', '')
            if 'cfl_' not in name:
                continue
            a = out.setdefault(name, {}).setdefault(ctr, [0, 0.0])
            a[0] += 1
            a[1] += float(r['Counter_Value'])
res = {'_method': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes); FETCH_SIZE x 2 (gfx950 correction), KB -> bytes'}
for k, v in sorted(out.items()):
    f = v.get('FETCH_SIZE', [0, 0.0]); w = v.get('WRITE_SIZE', [0, 0.0])
    fk = f[1] / f[0] if f[0] else 0.0
    wk = w[1] / w[0] if w[0] else 0.0
    res[k] = {'launches': f[0] or w[0], 'avg_fetch_kb_raw': round(fk, 1), 'avg_write_kb': round(wk, 1),
              'traffic_bytes': int(2 * fk * 1024 + wk * 1024)}
    if STEPS:
        res[k]['launches_per_step'] = round((f[0] or w[0]) / STEPS, 3)
print(json.dumps(res, indent=1))
