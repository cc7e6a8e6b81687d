#!/usr/bin/env python3
"""Whole-step HBM accounting of the bench step: bytes per hardware queue and per kernel family for EVERY kernel of the timed steps --
library kernels (MIOpen / CK / hipBLASLt / aten) included, which tools/pmc_summary.py leaves out -- from the two PMC passes of
tools/pmc_run.sh (FETCH_SIZE, WRITE_SIZE: separate rocprofv3 runs, never combined with another trace domain).

    python tools/step_bytes.py gpurun_out/pmc_<tag> [--after-nth cfl_adamp_pass3_kernel 1] [--step-ms 44.3] [--tflop 15.08]

FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 bytes: MI355X_MICROARCH.md, HBM section);
WRITE_SIZE as reported; both arrive in KB.  The doubling is exact for 16-byte-per-lane streaming reads (all hand-written kernels here,
checked against their algorithmic bytes: 1.04 x) and UNCALIBRATED for the library kernels' access patterns: their figure is an upper
estimate, and the floor derived from it is stated as such."""
import argparse, collections, csv, glob, json, os, re, sys

ap = argparse.ArgumentParser()
ap.add_argument('dir')
ap.add_argument('--after-nth', nargs=2, default=['cfl_adamp_pass3_kernel', '1'])
ap.add_argument('--step-ms', type=float, default=0.0, help='un-profiled wall time of the step, for the achieved whole-step rate')
ap.add_argument('--tflop', type=float, default=0.0, help='analytic FLOP of the step (bench line: mfu.tflop_per_step)')
ap.add_argument('--top', type=int, default=14)
a = ap.parse_args()
marker, nth = a.after_nth[0], int(a.after_nth[1])


def family(name):
    if 'cfl_bn_' in name:
        return 'cfl BatchNorm family'
    if 'cfl_gemm_bf16' in name:
        return 'cfl GEMM (1x1 data gradients, conv3 forward)'
    if 'cfl_adamp' in name or 'cfl_clip' in name or 'cfl_gradnorm' in name:
        return 'cfl clip + AdamP'
    if 'cfl_' in name:
        return 'cfl other (BERT glue, PIE, pair loss, pool, transposes)'
    if 'igemm_wrw' in name or ('batched_gemm' in name):
        return 'library weight gradients (MIOpen igemm_wrw, CK batched GEMM)'
    if 'conv_fwd' in name or 'igemm_fwd' in name or 'igemm_bwd' in name or 'conv_bwd_data' in name or 'naive_conv' in name:
        return 'library forward / data-gradient convolutions'
    if name.startswith('Cijk_') or name.startswith('Custom_Cijk_'):
        return 'hipBLASLt (BERT linears)'
    return 'other library / aten'


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n).split('(')[0].replace('void ', '')
    return n[:100]


per_ctr = {}
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    rows = []
    for f in glob.glob(os.path.join(a.dir, ctr, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') == ctr:
                rows.append((int(r.get('Dispatch_Id', 0)), r.get('Queue_Id', '?'), short(r['Kernel_Name']), float(r['Counter_Value'])))
    rows.sort()
    seen, start, ends = 0, None, []
    for i, (_, _, n, _) in enumerate(rows):
        if marker in n:
            seen += 1
            if seen == nth:
                start = i + 1
            elif seen > nth:
                ends.append(i + 1)
    if start is None or not ends:
        sys.exit('%s: marker kernel not found often enough' % ctr)
    per_ctr[ctr] = (rows[start:ends[-1]], len(ends))

steps = min(v[1] for v in per_ctr.values())
scale = {'FETCH_SIZE': 2 * 1024.0, 'WRITE_SIZE': 1024.0}
by_q = collections.defaultdict(lambda: [0.0, 0.0, 0])
by_f = collections.defaultdict(lambda: [0.0, 0.0, 0])
by_k = collections.defaultdict(lambda: [0.0, 0.0, 0])
for ci, ctr in enumerate(('FETCH_SIZE', 'WRITE_SIZE')):
    rows, n = per_ctr[ctr]
    for _, q, name, v in rows:
        b = v * scale[ctr] / n
        by_q[q][ci] += b
        by_f[family(name)][ci] += b
        by_k[name][ci] += b
        if ci == 0:
            by_q[q][2] += 1.0 / n
            by_f[family(name)][2] += 1.0 / n
            by_k[name][2] += 1.0 / n


def rec(v):
    return {'read_gb': round(v[0] / 1e9, 3), 'write_gb': round(v[1] / 1e9, 3), 'total_gb': round((v[0] + v[1]) / 1e9, 3),
            'launches_per_step': round(v[2], 1)}


total = sum(v[0] + v[1] for v in by_q.values())
out = {'_method': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes), timed steps only; FETCH_SIZE x 2 (gfx950), KB -> bytes; '
                  'library kernels\' read figure is uncalibrated (upper estimate)',
       'steps': steps,
       'bytes_per_step_gb': round(total / 1e9, 2),
       'by_queue': {q: rec(v) for q, v in sorted(by_q.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))},
       'by_family': {f: rec(v) for f, v in sorted(by_f.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))},
       'top_kernels': {k: rec(v) for k, v in sorted(by_k.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:a.top]},
       'floor_ms': {'hbm_at_6.3_TBps_achievable': round(total / 6.3e12 * 1e3, 2), 'hbm_at_8_TBps_spec': round(total / 8e12 * 1e3, 2)}}
if a.tflop:
    out['floor_ms']['mfma_bf16_dense_2.5_PFLOPs'] = round(a.tflop / 2500.0 * 1e3, 2)
    out['tflop_per_step'] = a.tflop
if a.step_ms:
    out['step_ms'] = a.step_ms
    out['achieved_whole_step_tbps'] = round(total / a.step_ms / 1e9, 3)
    out['frac_of_achievable_hbm'] = round(total / a.step_ms / 1e9 / 6.3, 3)
print(json.dumps(out, indent=1))
