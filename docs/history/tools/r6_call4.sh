#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call4; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
export MIOPEN_LOG_LEVEL=1
( time timeout 900 python -m pytest tests/test_gpu_x3conv.py -q ) > $OUT/x3_tests.log 2>&1; tail -n 4 $OUT/x3_tests.log
timeout 900 python tools/kernel_bench.py --cases x3conv > $OUT/x3conv_probe.jsonl 2> $OUT/x3conv_probe.err; tail -n 3 $OUT/x3conv_probe.err
python3 - <<'PY'
import json,os
for ln in open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/call4/x3conv_probe.jsonl'):
    if ln.startswith('{'):
        d=json.loads(ln); print(d['case'], {k:v for k,v in d.items() if k.endswith('fwd_us') and ('v2' in k or 'v0' in k) or 'x3_wgrad' in k})
PY
for HW in "28 128" "14 256"; do set -- $HW
  bash tools/pmc_sq.sh x3_$1 python $ROOT/tools/x3_one.py --hw $1 --c $2 > $OUT/sq_x3_$1.json 2>&1; cat $OUT/sq_x3_$1.json | tail -n 60
  bash tools/pmc_run.sh x3_$1 python $ROOT/tools/x3_one.py --hw $1 --c $2 > $OUT/pmc_x3_$1.log 2>&1; tail -n 30 $OUT/pmc_x3_$1.log
done
timeout 1500 python bench.py --config 2 --no-cpu-baseline --steps 10 --warmup 3 --round none > $OUT/config2_line.json 2> $OUT/config2.err
python3 -c "
import json
d=json.load(open('$OUT/config2_line.json'))
print({k:(v.get('graph') or v.get('eager') or {}).get('ms_per_step') for k,v in d['clients'].items()})
print({k:v for k,v in d['clients']['img']['hip_kernels'].items() if 'conv' in k})"
tail -n 3 $OUT/config2.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace_img -o img --output-format csv -- python $ROOT/bench.py --config 2 --no-cpu-baseline --steps 5 --warmup 2 --round none --only-kinds img > $OUT/trace_img.log 2>&1
python3 - <<'PY'
import csv,glob,os
root=os.environ.get('GRAFT_REPO_ROOT','.')
fs=glob.glob(root+'/gpurun_out/call4/trace_img/**/*kernel_stats.csv',recursive=True)
print(fs)
for f in fs:
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r['TotalDurationNs']))
    for r in rows[:40]: print(r['Name'][:110], r['Calls'], round(float(r['TotalDurationNs'])/1e6,2), round(float(r['AverageNs'])/1e3,1))
PY
