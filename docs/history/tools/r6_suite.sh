#!/bin/bash
# round 6: the whole GPU suite + smoke + the two bench lines (what the driver runs at round end), one gpurun call
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r6}
OUT=$ROOT/gpurun_out/suite_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
( time timeout 3000 python -m pytest tests -q -m gpu ) > $OUT/${TAG}_gputest.log 2>&1
tail -n 15 $OUT/${TAG}_gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -n 2 $OUT/smoke.log
python bench.py > $OUT/${TAG}_bench_line_driver.json 2> $OUT/bench.err
python3 -c "
import json
d=json.load(open('$OUT/${TAG}_bench_line_driver.json')); r=d['roofline']
print(d['ms_per_step'], d['value'], r['frac'], r['avg_launch_us'], r['traffic'], d['cpu_baseline'], (d.get('extra') or {}).keys())"
tail -n 3 $OUT/bench.err
timeout 1500 python bench.py --config 2 --no-cpu-baseline > $OUT/${TAG}_config2_line.json 2> $OUT/config2.err
python3 -c "
import json
d=json.load(open('$OUT/${TAG}_config2_line.json'))
print({k:(v.get('graph') or v.get('eager') or {}).get('ms_per_step') for k,v in d['clients'].items()})
print(d['round']['ms_per_public_batch'], d['round']['phases_s_rank0'])"
tail -n 3 $OUT/config2.err
