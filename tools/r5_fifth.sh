#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_fifth
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_bnorm.py > $OUT/bnorm_tests.log 2>&1; tail -4 $OUT/bnorm_tests.log | cut -c1-400
timeout 900 python tools/train_outcome_probe.py > $OUT/outcome.jsonl 2> $OUT/outcome.err; cat $OUT/outcome.jsonl | cut -c1-500
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace_bench -o bench --output-format csv -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-recall --no-alone --no-mfu > $OUT/trace_bench.log 2>&1
T=$(ls $OUT/trace_bench/*kernel_trace.csv $OUT/trace_bench/*/*kernel_trace.csv 2>/dev/null | head -1)
head -1 $T > $OUT/trace_header.txt
python3 $ROOT/tools/trace_streams.py $T > $OUT/streams.json 2> $OUT/streams.err
python3 $ROOT/tools/trace_stats.py $T > $OUT/bench_kernel_stats.csv
head -c 6000 $OUT/streams.json
rm -rf $OUT/trace_bench
cd $ROOT
bash tools/record_find_db.sh > $OUT/record.log 2>&1; tail -12 $OUT/record.log
