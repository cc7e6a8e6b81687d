#!/usr/bin/env python3
"""Which stream is the step's critical path?  From a rocprofv3 --kernel-trace csv of a bench run: per hardware queue, over the
timed steps only (everything before the n-th launch of a marker kernel is dropped), the busy time (union of kernel intervals),
the summed kernel time and the top kernels -- the queue whose busy time is closest to the wall time carries the step.
    python tools/trace_streams.py <kernel_trace.csv> [--after-nth cfl_adamp_pass3_kernel 2] [--top 12]"""
import argparse, collections, csv, json, re, sys
ap = argparse.ArgumentParser()
ap.add_argument('trace')
ap.add_argument('--after-nth', nargs=2, default=['cfl_adamp_pass3_kernel', '2'])
ap.add_argument('--top', type=int, default=12)
ap.add_argument('--gaps', type=int, default=0, help='list the n largest idle gaps per step of every queue with the kernels around them')
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marker, nth = a.after_nth[0], int(a.after_nth[1])
seen, t0 = 0, None
steps = 0
for r in rows:
    if marker in r['Kernel_Name']:
        seen += 1
        if seen == nth:
            t0 = int(r['End_Timestamp'])
        elif seen > nth:
            steps += 1
            t1 = int(r['End_Timestamp'])
if t0 is None or steps == 0:
    sys.exit('marker kernel not found often enough')
qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else ('Stream_Id' if 'Stream_Id' in rows[0] else None)
per = collections.defaultdict(list)
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s < t0 or e > t1:
        continue
    per[r.get(qkey, '?') if qkey else '?'].append((s, e, re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace('void ', '')[:90]))
wall = (t1 - t0) / steps / 1e6
out = {'steps': steps, 'wall_ms_per_step': round(wall, 3), 'queue_key': qkey, 'queues': {}}
for q, iv in sorted(per.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    iv.sort()
    busy, cur_s, cur_e = 0, None, None
    gaps = []
    gap_list = []
    last_name = None
    for s, e, nm in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
                gaps.append(s - cur_e)
                gap_list.append((s - cur_e, last_name, nm, cur_e - t0))
            cur_s, cur_e = s, e
            last_name = nm
        else:
            if e > cur_e:
                cur_e, last_name = e, nm
    busy += (cur_e - cur_s) if cur_e is not None else 0
    agg = collections.Counter()
    cnt = collections.Counter()
    for s, e, n in iv:
        agg[n] += e - s
        cnt[n] += 1
    out['queues'][str(q)] = {'launches_per_step': round(len(iv) / steps, 1), 'busy_ms_per_step': round(busy / steps / 1e6, 3),
                             'kernel_ms_per_step': round(sum(agg.values()) / steps / 1e6, 3),
                             # idle intervals between consecutive busy intervals of this queue: how much of the step the queue waits
                             # (for the host, for another queue's event, for the dispatch of a dependent kernel)
                             'gaps': {'per_step': round(len(gaps) / steps, 1), 'idle_ms_per_step': round(sum(gaps) / steps / 1e6, 3),
                                      'median_us': round(sorted(gaps)[len(gaps) // 2] / 1e3, 2) if gaps else 0,
                                      'idle_ms_in_gaps_over_20us': round(sum(g for g in gaps if g > 20000) / steps / 1e6, 3),
                                      'n_over_20us_per_step': round(sum(1 for g in gaps if g > 20000) / steps, 1)},
                             'top': [{'kernel': n, 'ms_per_step': round(t / steps / 1e6, 3), 'launches_per_step': round(cnt[n] / steps, 1),
                                      'avg_us': round(t / cnt[n] / 1e3, 1)} for n, t in agg.most_common(a.top)]}
    if a.gaps:
        gap_list.sort(reverse=True)
        out['queues'][str(q)]['largest_gaps'] = [{'us': round(g / 1e3, 1), 'after': b[:60], 'before': n[:60],
                                                  'at_ms_into_window': round(at / 1e6, 2)} for g, b, n, at in gap_list[:a.gaps * steps]]
print(json.dumps(out, indent=1))
