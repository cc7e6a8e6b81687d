#!/usr/bin/env python3
"""Per-queue busy time and idle gaps of the timed steps of a rocprofv3 --kernel-trace csv (bench.py run).
   python tools/queue_busy.py <kernel_trace.csv> [--after-nth cfl_adamp_pass3_kernel 2]"""
import argparse, collections, csv, sys
ap = argparse.ArgumentParser()
ap.add_argument('trace')
ap.add_argument('--after-nth', nargs=2, default=['cfl_adamp_pass3_kernel', '2'])
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marker, nth = a.after_nth[0], int(a.after_nth[1])
seen, t0 = 0, None
steps_end = []
for r in rows:
    if marker in r['Kernel_Name']:
        seen += 1
        if seen == nth:
            t0 = int(r['End_Timestamp'])
        if seen > nth:
            steps_end.append(int(r['End_Timestamp']))
if t0 is None or not steps_end:
    sys.exit('marker kernel not found often enough')
t1 = steps_end[-1]
nsteps = len(steps_end)
qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else ('Stream_Id' if 'Stream_Id' in rows[0] else None)
per = collections.defaultdict(list)
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s >= t0 and e <= t1:
        per[r[qkey] if qkey else '0'].append((s, e, r['Kernel_Name'][:60]))
print('window %.2f ms = %d steps of %.2f ms' % ((t1 - t0) / 1e6, nsteps, (t1 - t0) / 1e6 / nsteps))
allk = sorted(k for v in per.values() for k in v)
# union busy time
busy, cur_s, cur_e = 0, None, None
for s, e, _ in allk:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('any queue busy: %.2f ms/step (idle %.2f ms/step)' % (busy / 1e6 / nsteps, ((t1 - t0) - busy) / 1e6 / nsteps))
for q, ks in sorted(per.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    tot = sum(e - s for s, e, _ in ks)
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    small = sum(g for g in gaps if 0 < g <= 5000)
    big = sum(g for g in gaps if g > 5000)
    print('queue %s: %5d kernels/step, busy %.2f ms/step, gaps<=5us %.2f ms/step, gaps>5us %.2f ms/step' %
          (q, len(ks) // nsteps, tot / 1e6 / nsteps, small / 1e6 / nsteps, big / 1e6 / nsteps))
