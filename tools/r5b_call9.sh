#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5b10
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
timeout 200 python tools/federation_step_trace.py --mini-first 1 2> $OUT/t1.err | tee $OUT/trace.jsonl | cut -c1-700
timeout 200 python tools/federation_step_trace.py --measure-first 1 2> $OUT/t2.err | tee -a $OUT/trace.jsonl | cut -c1-700
tail -2 $OUT/t1.err $OUT/t2.err | cut -c1-200
