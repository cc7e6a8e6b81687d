#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_suite
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
( time python -m pytest tests -m gpu -x -q --durations=15 ) > $OUT/gputest.log 2>&1
tail -40 $OUT/gputest.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
