#!/bin/bash
# round 5, GPU call: sliced-BN parity + A/B, then what r5_first.sh runs
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_second
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_bnorm.py > $OUT/bnorm_tests.log 2>&1
tail -15 $OUT/bnorm_tests.log
timeout 900 python tools/ab_step.py --knob bnslice --rounds 6 > $OUT/ab_bnslice.json 2> $OUT/ab.err
cat $OUT/ab_bnslice.json
bash tools/r5_first.sh
