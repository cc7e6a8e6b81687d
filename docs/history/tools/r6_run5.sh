#!/bin/bash
# round 6, GPU call 5: direct finish of the bank pass (parity + timing), why bench.py reads 46.0 ms where ab_step reads 42.3, outcome-task calibration
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_run5
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
( timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "a34 or a3_ or abi or stub" ) > $OUT/test_a34.log 2>&1
tail -n 12 $OUT/test_a34.log
( timeout 1500 python -m pytest tests/test_gpu_framework.py tests/test_gpu_configs.py -q -m gpu -k "contrast or client or round" ) > $OUT/test_clients.log 2>&1
tail -n 8 $OUT/test_clients.log
timeout 600 python tools/kernel_bench.py --cases a3 > $OUT/r6_a3_kernel_bench.jsonl 2> $OUT/kb.err
cat $OUT/r6_a3_kernel_bench.jsonl; tail -n 3 $OUT/kb.err
B="python bench.py --no-cpu-baseline --no-client-steps --no-mfu"
$B --steps 20 --warmup 5 > $OUT/bench_w5.json 2>> $OUT/bench.err
$B --steps 20 --warmup 20 > $OUT/bench_w20.json 2>> $OUT/bench.err
$B --steps 60 --warmup 20 > $OUT/bench_w20_s60.json 2>> $OUT/bench.err
CFL_NO_WGRAD1=1 $B --steps 20 --warmup 5 > $OUT/bench_now1.json 2>> $OUT/bench.err
CFL_NO_WGRAD1=1 CFL_NO_WGRAD3=1 $B --steps 20 --warmup 5 > $OUT/bench_now13.json 2>> $OUT/bench.err
BENCH_NO_PROF=1 $B --steps 20 --warmup 5 > $OUT/bench_noprof.json 2>> $OUT/bench.err
for f in w5 w20 w20_s60 now1 now13 noprof; do python3 -c "
import json,sys
d=json.load(open('$OUT/bench_$f.json')); r=d.get('roofline') or {}
print('$f', d['ms_per_step'], r.get('avg_launch_us'), r.get('frac'))"; done
tail -n 3 $OUT/bench.err
for CFG in "--steps 400" "--steps 700"; do
  timeout 1200 python tools/train_outcome_probe.py --n-id 200 --caption-swap 0.25 --seeds 3 $CFG >> $OUT/r6_outcome_calibration2.jsonl 2>> $OUT/outcome.err
done
python3 -c "
import json
for l in open('$OUT/r6_outcome_calibration2.jsonl'):
    d=json.loads(l); print(d['steps'], d['seed'], 'fp32' if d['fp32'] else 'bf16', d['i2t_r1'], d['t2i_r1'], d['losses'][-3:], d['train_s'])"
tail -n 3 $OUT/outcome.err
