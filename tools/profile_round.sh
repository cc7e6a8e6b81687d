#!/bin/bash
# Round evidence in one gpurun call (run from the repo root):  bash tools/profile_round.sh r2
#   <tag>_bench_line.json          python bench.py (default flags = the driver's command line)
#   <tag>_bench_fp32_line.json     the same with fp32 trunks (companion number: north star quotes fp32 tolerances)
#   <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of a short bench run
#   <tag>_a3_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the client contrast step (A3 + A4)
#   <tag>_kernel_bench.jsonl       tools/kernel_bench.py at the SURVEY 8(d) shapes
#   <tag>_wall_a3.jsonl            wall / host / HIP-graph time of the client contrast step
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd $ROOT
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err
python bench.py --steps 10 --warmup 3 --dtype fp32 --no-cpu-baseline --no-recall > $OUT/${TAG}_bench_fp32_line.json 2>> $OUT/bench.err
python tools/kernel_bench.py --cases a1,a3,a5,a2,a6,f4,pool,gemm16,opt > $OUT/${TAG}_kernel_bench.jsonl 2> $OUT/kb.err
python tools/wall_a3.py > $OUT/${TAG}_wall_a3.jsonl 2>> $OUT/kb.err
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace_a3 $OUT/trace_bench
rocprofv3 --kernel-trace --stats -d $OUT/trace_a3 -o a3 --output-format csv -- python $ROOT/tools/kernel_bench.py --cases a3one > $OUT/trace_a3.log 2>&1
cp $OUT/trace_a3/a3_kernel_stats.csv $OUT/${TAG}_a3_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $OUT/trace_bench -o bench --output-format csv -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-recall --no-alone > $OUT/trace_bench.log 2>&1
cp $OUT/trace_bench/bench_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
rm -rf $OUT/trace_a3 $OUT/trace_bench
ls -la $OUT
