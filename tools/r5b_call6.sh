#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5b7
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
timeout 400 python tools/federation_step_trace.py 2> $OUT/t.err | tee $OUT/r5_federation_step_trace.jsonl | cut -c1-1500; tail -2 $OUT/t.err | cut -c1-200
timeout 400 python tools/federation_step_trace.py --sync-every 1 2> $OUT/t2.err | tee -a $OUT/r5_federation_step_trace.jsonl | cut -c1-1500
