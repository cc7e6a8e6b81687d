#!/bin/bash
# the final lines of the round (bench first: a fresh box), then the whole GPU suite + smoke
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_suite
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r5_bench_line_final.json 2> $OUT/bench.err
timeout 1200 python bench.py --config 2 --steps 30 --warmup 5 > $OUT/r5_config2_line.json 2> $OUT/c2.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_suite/r5_bench_line_final.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['alone']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'][:44], d['mfu']['mfu'], d['cpu_baseline']['value'])
d=json.loads(open('gpurun_out/r5_suite/r5_config2_line.json').read().strip().splitlines()[-1])
print('config2', d['value'], {k:(v['eager']['ms_per_step'], v.get('graph') and v['graph']['ms_per_step'], v['a3a4_kernels_us_per_step'], v['a3a4_share_of_step']['product_path'], v['hand_written_kernels_us_per_step']) for k,v in d['clients'].items()}, d['round']['miniature_warm_up_round_s'], d['round']['phases_s_rank0'], d['round']['ms_per_public_batch'], d['full_M']['con_w_ms_per_client'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['cpu_baseline']['value'])
PY
( time python -m pytest tests -m gpu -q --durations=6 ) > $OUT/gputest.log 2>&1
tail -14 $OUT/gputest.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
# every kernel of the three clients' contrast steps at the final code
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace_client -o cl --output-format csv -- python $ROOT/bench.py --config 2 --round none --steps 20 --warmup 3 --no-cpu-baseline > $OUT/trace_client.log 2>&1
python3 - $OUT <<'PY'
import csv, sys, glob
f = (glob.glob(sys.argv[1]+'/trace_client/*kernel_stats.csv')+glob.glob(sys.argv[1]+'/trace_client/*/*kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(f)))
w = csv.writer(open(sys.argv[1]+'/r5_client_step_kernel_stats.csv','w'))
w.writerow(['Name','Calls','TotalDurationNs','AverageNs','Percentage','MinNs','MaxNs'])
for r in rows: w.writerow([r['Name'][:140], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
PY
rm -rf $OUT/trace_client
head -12 $OUT/r5_client_step_kernel_stats.csv | cut -c1-160
