#!/bin/bash
# HBM traffic per kernel: two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains),
# summarised per kernel.  usage (via gpurun, from the repo root):  [PMC_STEPS=<warm-up + timed steps>] bash tools/pmc_run.sh <tag> <command ...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/$C -o pmc --output-format csv -- "$@" > $OUT/$C.log 2>&1
done
python3 $ROOT/tools/pmc_summary.py $OUT $PMC_STEPS > $OUT/summary.json
cat $OUT/summary.json
