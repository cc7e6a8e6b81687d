#!/bin/bash
# round 6, GPU call 4: wgrad3x3 v3 + wgrad1x1 (parity, stand-alone timing, decomposition, in-step A/Bs), outcome-test calibration
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_run4
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
( timeout 1200 python -m pytest tests/test_gpu_wgrad.py -q -m gpu ) > $OUT/test_wgrad.log 2>&1
tail -n 25 $OUT/test_wgrad.log
timeout 900 python tools/kernel_bench.py --cases wgrad3,w3dbg,wgrad1 > $OUT/r6_wgrad_kernel_bench.jsonl 2> $OUT/kb.err
cat $OUT/r6_wgrad_kernel_bench.jsonl; tail -n 3 $OUT/kb.err
for K in wgrad3 wgrad1; do
  timeout 900 python tools/ab_step.py --knob $K --rounds 6 > $OUT/r6_ab_$K.json 2>> $OUT/ab.err
  cat $OUT/r6_ab_$K.json
done
tail -n 3 $OUT/ab.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-client-steps > $OUT/r6_bench_line_wgrad.json 2> $OUT/bench.err
cut -c1-400 $OUT/r6_bench_line_wgrad.json
for CFG in "--n-id 1000 --noise 0.5 --steps 300" "--n-id 1000 --noise 0.5 --steps 500" "--n-id 500 --noise 0.6 --steps 300"; do
  timeout 900 python tools/train_outcome_probe.py $CFG --n-eval 500 >> $OUT/r6_outcome_calibration.jsonl 2>> $OUT/outcome.err
done
cut -c1-330 $OUT/r6_outcome_calibration.jsonl; tail -n 3 $OUT/outcome.err
