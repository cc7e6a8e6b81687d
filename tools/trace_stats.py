#!/usr/bin/env python3
"""Per-kernel statistics of the TIMED steps of a rocprofv3 --kernel-trace csv (bench.py run): everything before the n-th
launch of a marker kernel (default: the last kernel of an optimizer step) is dropped, so library search / warm-up kernels do
not pollute the shares.   python tools/trace_stats.py <kernel_trace.csv> [--after-nth cfl_adamp_pass3_kernel 2] > stats.csv"""
import argparse, collections, csv, re, sys
ap = argparse.ArgumentParser()
ap.add_argument('trace')
ap.add_argument('--after-nth', nargs=2, default=['cfl_adamp_pass3_kernel', '2'])
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marker, nth = a.after_nth[0], int(a.after_nth[1])
seen, t0 = 0, None
for r in rows:
    if marker in r['Kernel_Name']:
        seen += 1
        if seen == nth:
            t0 = int(r['End_Timestamp'])
            break
if t0 is None:
    sys.exit('marker kernel not found %d times' % nth)
agg = collections.OrderedDict()
for r in rows:
    if int(r['Start_Timestamp']) < t0:
        continue
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = name.split('(')[0].replace('void ', '')[:120]
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    s = agg.setdefault(name, [0, 0, 1 << 62, 0])
    s[0] += 1; s[1] += d; s[2] = min(s[2], d); s[3] = max(s[3], d)
tot = sum(v[1] for v in agg.values())
w = csv.writer(sys.stdout)
w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    w.writerow([k, v[0], v[1], round(v[1] / v[0], 1), round(100.0 * v[1] / tot, 3), v[2], v[3]])
