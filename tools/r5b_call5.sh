#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5b6
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
for b in 128 256; do timeout 400 python tools/server_graph_ab.py --batch $b 2> $OUT/ab_$b.err | tee -a $OUT/r5_server_graph_ab.jsonl | cut -c1-900; tail -2 $OUT/ab_$b.err | cut -c1-200; done
