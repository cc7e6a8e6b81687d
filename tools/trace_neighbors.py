#!/usr/bin/env python3
"""Who launches the small library kernels of a step?  From a rocprofv3 --kernel-trace csv: for every launch whose name matches
--match, the queue it ran on and the kernels right before / after it on that queue, counted over the timed steps (after the n-th
marker kernel).    python tools/trace_neighbors.py <kernel_trace.csv> --match copyBuffer [--after-nth cfl_adamp_pass3_kernel 2]"""
import argparse, collections, csv, json, re
ap = argparse.ArgumentParser()
ap.add_argument('trace')
ap.add_argument('--match', default='copyBuffer')
ap.add_argument('--after-nth', nargs=2, default=['cfl_adamp_pass3_kernel', '2'])
ap.add_argument('--top', type=int, default=25)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marker, nth = a.after_nth[0], int(a.after_nth[1])
seen, t0, t1, steps = 0, None, None, 0
for r in rows:
    if marker in r['Kernel_Name']:
        seen += 1
        if seen == nth:
            t0 = int(r['End_Timestamp'])
        elif seen > nth:
            steps += 1
            t1 = int(r['End_Timestamp'])
qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else 'Stream_Id'
short = lambda n: re.sub(r'\(anonymous namespace\)::', '', n).split('(')[0].replace('void ', '')[:70]
per = collections.defaultdict(list)
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if t0 is None or s < t0 or e > t1:
        continue
    per[r[qkey]].append((s, e, short(r['Kernel_Name'])))
out = {'steps': steps, 'match': a.match, 'queues': {}}
for q, iv in per.items():
    iv.sort()
    pairs, n, us = collections.Counter(), 0, 0.0
    for i, (s, e, k) in enumerate(iv):
        if a.match in k:
            n += 1
            us += (e - s) / 1e3
            pairs[(iv[i - 1][2] if i else '-', iv[i + 1][2] if i + 1 < len(iv) else '-')] += 1
    if n:
        out['queues'][q] = {'per_step': round(n / steps, 1), 'us_per_step': round(us / steps, 1),
                            'neighbours': [{'before': b, 'after': c, 'per_step': round(v / steps, 1)} for (b, c), v in pairs.most_common(a.top)]}
print(json.dumps(out, indent=1))
