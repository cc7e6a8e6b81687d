#!/bin/bash
# round 6, call 12: where the step's ~145 device copies and ~150 fills come from (kernel trace + neighbours), configs[4] line at this round's code
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call12; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace_bench -o bench --output-format csv -- python3 $ROOT/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-client-steps --no-recall --no-mfu --no-alone > $OUT/trace_bench.log 2>&1
cd $ROOT
T=$(ls $OUT/trace_bench/*kernel_trace.csv | head -n 1)
python3 tools/trace_neighbors.py $T --match copyBuffer > $OUT/copies.json; head -c 3000 $OUT/copies.json
python3 tools/trace_neighbors.py $T --match FillFunctor > $OUT/fills.json; head -c 2500 $OUT/fills.json
python3 tools/trace_neighbors.py $T --match elementwise > $OUT/elementwise.json; head -c 2500 $OUT/elementwise.json
rm -rf $OUT/trace_bench
timeout 900 python tools/config4_bench.py --batch 256 > $OUT/r6_config4_b256_line.json 2> $OUT/c4.err; cat $OUT/r6_config4_b256_line.json; tail -n 2 $OUT/c4.err
