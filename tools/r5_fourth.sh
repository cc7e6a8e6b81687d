#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_fourth
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_bnorm.py > $OUT/bnorm_tests.log 2>&1; tail -12 $OUT/bnorm_tests.log | cut -c1-600
timeout 900 python tools/ab_step.py --knob wgfuse --rounds 6 > $OUT/ab_wgfuse.json 2> $OUT/ab.err
cat $OUT/ab_wgfuse.json; tail -3 $OUT/ab.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall > $OUT/bench_line.json 2> $OUT/bench.err
cat $OUT/bench_line.json | cut -c1-1200
