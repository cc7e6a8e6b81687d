#!/bin/bash
# round 6, GPU call 7: does the main stream wait for the weight-gradient stream at the end of the backward pass (HIP events), flush
# policies, the text-tower / both-sides bounds, the bench line at the new defaults
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_run7
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
for F in 0 1 2 6; do timeout 600 python tools/tail_probe.py --flush $F >> $OUT/r6_tail_probe.jsonl 2>> $OUT/tail.err; done
cat $OUT/r6_tail_probe.jsonl; tail -n 3 $OUT/tail.err
for K in flush1 flush2 text sides; do
  timeout 600 python tools/ab_step.py --knob $K --rounds 4 > $OUT/r6_ab_$K.json 2>> $OUT/ab.err
  cat $OUT/r6_ab_$K.json
done
tail -n 3 $OUT/ab.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-client-steps > $OUT/r6_bench_line_w1hw28.json 2>> $OUT/bench.err
python3 -c "
import json
d=json.load(open('$OUT/r6_bench_line_w1hw28.json')); r=d['roofline']
print(d['ms_per_step'], d['value'], r['avg_launch_us'], r['frac'], r['traffic'], r.get('alone'))"
tail -n 3 $OUT/bench.err
