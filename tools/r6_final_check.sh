#!/bin/bash
# round 6, last check at the committed code: the driver's three commands (GPU suite, smoke, bench)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/final_check; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
export MIOPEN_LOG_LEVEL=1
( time timeout 2400 python -m pytest tests -q -m gpu ) > $OUT/r6_gputest.log 2>&1
tail -n 6 $OUT/r6_gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -n 2 $OUT/smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r6_bench_line_driver.json 2> $OUT/bench.err
python3 -c "
import json
d=json.load(open('$OUT/r6_bench_line_driver.json')); r=d['roofline']
print('bench', d['ms_per_step'], d['value'], r['frac'], r['avg_launch_us'], d['cpu_baseline']['value'], {k:(v.get('ms_per_step') if isinstance(v,dict) else None) for k,v in d['extra']['client_steps'].items() if k in ('image','text','multi_modal')}); print(json.dumps(d['extra'].get('hot_kernels')))"
tail -n 2 $OUT/bench.err
