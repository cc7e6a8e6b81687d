#!/bin/bash
# round 6, call 15: full GPU suite + smoke + the driver's bench command at the code of this session (con_w wide kernels, ViT glue)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call15; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
export MIOPEN_LOG_LEVEL=1
( time timeout 2400 python -m pytest tests -q -m gpu -x ) > $OUT/r6_gputest.log 2>&1
tail -n 8 $OUT/r6_gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -n 2 $OUT/smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r6_bench_line_driver.json 2> $OUT/bench.err
python3 -c "
import json
d=json.load(open('$OUT/r6_bench_line_driver.json')); r=d['roofline']
print('bench', d['ms_per_step'], d['value'], r['frac'], r['avg_launch_us'], d['cpu_baseline']['value'], {k:(v.get('ms_per_step') if isinstance(v,dict) else None) for k,v in d['extra']['client_steps'].items() if k in ('image','text','multi_modal')})"
tail -n 2 $OUT/bench.err
