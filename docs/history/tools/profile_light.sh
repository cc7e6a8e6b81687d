#!/bin/bash
# The cheap part of tools/profile_round.sh (no fp32 bench, no library GEMM comparisons):  bash tools/profile_light.sh r2b
#   <tag>_bench_line.json          python bench.py --steps 20 --warmup 5 (the driver's command line)
#   <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace of a short bench run, timed steps only (tools/trace_stats.py)
#   <tag>_kernel_bench.jsonl       tools/kernel_bench.py --cases a1,a3,a5,a2,a6,pool,bn
TAG=${1:-r2b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd $ROOT
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err
python tools/kernel_bench.py --cases a1,a3,a5,a2,a6,pool,bn > $OUT/${TAG}_kernel_bench.jsonl 2> $OUT/kb.err
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace_bench
rocprofv3 --kernel-trace -d $OUT/trace_bench -o bench --output-format csv -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-recall --no-alone > $OUT/trace_bench.log 2>&1
python $ROOT/tools/trace_stats.py $(ls $OUT/trace_bench/*kernel_trace.csv $OUT/trace_bench/*/*kernel_trace.csv 2>/dev/null | head -1) > $OUT/${TAG}_bench_kernel_stats.csv
rm -rf $OUT/trace_bench
ls -la $OUT
