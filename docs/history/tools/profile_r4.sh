#!/bin/bash
# Round-4 evidence in one gpurun call (run from the repo root):  bash tools/profile_r4.sh [tag]
#   <tag>_bench_line.json            python bench.py --steps 20 --warmup 5 (the driver's command line: no pre-warm child, CPU baseline at batch 256)
#   <tag>_bench_line_again.json      the same, GPU part only (second process on the box)
#   <tag>_config3_line.json          python bench.py --config 3 (512 pairs per GPU: the N = 1 reference of the N = 4096 / 8-GPU config)
#   <tag>_loop_bench.jsonl           tools/loop_bench.py: TrainerEngine.train(loader) from host batches vs the resident step (pinned, pageable)
#   <tag>_wall_a3.jsonl              tools/wall_a3.py: client contrast step eager vs HIP graph
#   <tag>_kernel_bench.jsonl         tools/kernel_bench.py at the SURVEY 8(d) shapes (A1 image mode, con_w ring, ...)
#   <tag>_bench_kernel_stats.csv     rocprofv3 --kernel-trace of a short bench run, timed steps only (tools/trace_stats.py)
#   <tag>_pmc_bench_traffic.json     HBM traffic per hand-written kernel of the bench step (separate --pmc passes, launches_per_step)
#   <tag>_a1_kernel_stats.csv        rocprofv3 --kernel-trace --stats of the pair loss at N = 4096, D = 512 (image mode)
#   <tag>_a3one_kernel_stats.csv     ... of the client contrast step, B = 128, M = 50 000, D = 256
#   <tag>_pmc_a1.json                HBM traffic of the pair-loss kernels at N = 4096
#   <tag>_config4_line.json          BASELINE configs[4] at full encoder size (tools/config4_bench.py)
TAG=${1:-r4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-alone > $OUT/${TAG}_bench_line_again.json 2>> $OUT/bench.err
python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-recall --no-alone > $OUT/${TAG}_config3_line.json 2>> $OUT/bench.err
python tools/loop_bench.py 2>> $OUT/bench.err | tail -1 > $OUT/${TAG}_loop_bench.jsonl
python tools/loop_bench.py --pinned 0 --rounds 1 2>> $OUT/bench.err | tail -1 >> $OUT/${TAG}_loop_bench.jsonl
python tools/wall_a3.py 2>> $OUT/bench.err | tail -3 > $OUT/${TAG}_wall_a3.jsonl
python tools/kernel_bench.py --cases a1,a3,a5,a2,a6,f4,pool,gemm16,dgrad16,fwdstats16,opt > $OUT/${TAG}_kernel_bench.jsonl 2> $OUT/kb.err
python tools/config4_bench.py > $OUT/${TAG}_config4_line.json 2> $OUT/c4.err
python tools/config4_bench.py --batch 256 > $OUT/${TAG}_config4_b256_line.json 2>> $OUT/c4.err
PMC_STEPS=3 bash tools/pmc_run.sh bench python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-recall --no-alone --no-mfu > /dev/null 2>&1
cp $ROOT/gpurun_out/pmc_bench/summary.json $OUT/${TAG}_pmc_bench_traffic.json
bash tools/pmc_run.sh a1 python $ROOT/tools/kernel_bench.py --cases a1big > /dev/null 2>&1
cp $ROOT/gpurun_out/pmc_a1/summary.json $OUT/${TAG}_pmc_a1.json
cd /tmp && export TMPDIR=/tmp
cat > $OUT/pick.py <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'cfl_' in r['Name']]
w = csv.writer(sys.stdout)
w.writerow(['Name', 'Calls', 'AverageNs', 'MinNs', 'MaxNs'])
for r in rows:
    w.writerow([r['Name'].split('(float')[0].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', ''), r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs']])
PY
rocprofv3 --kernel-trace --stats -d $OUT/trace_a1 -o a1 --output-format csv -- python $ROOT/tools/kernel_bench.py --cases a1big > $OUT/trace_a1.log 2>&1
python3 $OUT/pick.py $OUT/trace_a1/a1_kernel_stats.csv > $OUT/${TAG}_a1_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $OUT/trace_a3one -o a3 --output-format csv -- python $ROOT/tools/kernel_bench.py --cases a3one > $OUT/trace_a3one.log 2>&1
python3 $OUT/pick.py $OUT/trace_a3one/a3_kernel_stats.csv > $OUT/${TAG}_a3one_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $OUT/trace_a5 -o a5 --output-format csv -- python $ROOT/tools/kernel_bench.py --cases a5 > $OUT/trace_a5.log 2>&1
python3 $OUT/pick.py $OUT/trace_a5/a5_kernel_stats.csv > $OUT/${TAG}_a5_kernel_stats.csv
rocprofv3 --kernel-trace -d $OUT/trace_bench -o bench --output-format csv -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-recall --no-alone > $OUT/trace_bench.log 2>&1
python3 $ROOT/tools/trace_stats.py $(ls $OUT/trace_bench/*kernel_trace.csv $OUT/trace_bench/*/*kernel_trace.csv 2>/dev/null | head -1) > $OUT/${TAG}_bench_kernel_stats.csv
rm -rf $OUT/trace_a1 $OUT/trace_a3one $OUT/trace_a5 $OUT/trace_bench
ls -la $OUT
