#!/bin/bash
# rocprofv3 kernel trace of the fused client-contrast step (A3 + A4) at B=128, M=50000, D=256.  Run via gpurun from the repo root.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_a3
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o a3 --output-format csv -- python $GRAFT_REPO_ROOT/tools/kernel_bench.py --cases a3one > $OUT/run.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -25 {} | cut -c1-200'
