#!/bin/bash
# round 6, GPU call 8: packed BERT tower (parity, in-step A/B, bench line), wgrad tests after the gate change
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_run8
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
( timeout 900 python -m pytest tests/test_gpu_bert.py -q -m gpu ) > $OUT/test_bert.log 2>&1
tail -n 15 $OUT/test_bert.log
timeout 600 python tools/ab_step.py --knob bertpack --rounds 4 > $OUT/r6_ab_bertpack.json 2>> $OUT/ab.err
cat $OUT/r6_ab_bertpack.json; tail -n 3 $OUT/ab.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-client-steps > $OUT/r6_bench_line_packed.json 2>> $OUT/bench.err
python3 -c "
import json
d=json.load(open('$OUT/r6_bench_line_packed.json')); r=d['roofline']
print(d['ms_per_step'], d['value'], r['avg_launch_us'], r['frac'], d['config']['text_tokens'], d['mfu']['mfu'], d['recall_1'])"
tail -n 3 $OUT/bench.err
( timeout 1500 python -m pytest tests/test_gpu_framework.py -q -m gpu -x -k "not outcome" ) > $OUT/test_framework.log 2>&1
tail -n 6 $OUT/test_framework.log
