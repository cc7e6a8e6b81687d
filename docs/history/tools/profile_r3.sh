#!/bin/bash
# Round-3 evidence in one gpurun call (run from the repo root):  bash tools/profile_r3.sh [tag]
#   <tag>_bench_line.json            python bench.py --steps 20 --warmup 5 (the driver's command line)
#   <tag>_bench_nojoin_line.json     the same with CFL_NO_JOIN_FUSE=1 (same-box A/B of the fused gradient join)
#   <tag>_bench_gloo2_line.json      python bench.py --gpus 2 --backend gloo (self-launched ranks; smoke mode of the N > 1 path)
#   <tag>_ab_join.json, <tag>_ab_bres.json   same-process A/B (tools/ab_step.py) of the gradient-join fusion / the B-resident GEMM
#   <tag>_step_jitter.json           per-step wall times of 80 consecutive steps (tools/step_jitter.py)
#   <tag>_bench_kernel_stats.csv     rocprofv3 --kernel-trace of a short bench run, timed steps only (tools/trace_stats.py)
#   <tag>_pmc_bench_traffic.json     HBM traffic per hand-written kernel of the bench step (separate --pmc passes, launches_per_step)
#   <tag>_kernel_bench.jsonl         tools/kernel_bench.py at the SURVEY 8(d) shapes
#   <tag>_a3one_kernel_stats.csv     rocprofv3 --kernel-trace --stats of the client contrast step, B = 128, M = 50 000, D = 256 (+ the round-2 path, same box)
#   <tag>_a3_kernel_stats.csv        the same at D = 256 / 512 / 768 (kernel instances tell the shapes apart)
#   <tag>_pmc_a3.json                HBM traffic of the bank pass / finish kernel (B = 128, M = 50 000, D = 256)
#   <tag>_sq_a3.json                 SQ / LDS / MFMA counters of the same
#   <tag>_config4_line.json, <tag>_config4_kernel_stats.csv   BASELINE configs[4] at full encoder size (tools/config4_bench.py)
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err
CFL_NO_JOIN_FUSE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-alone > $OUT/${TAG}_bench_nojoin_line.json 2>> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-alone > $OUT/${TAG}_bench_line_again.json 2>> $OUT/bench.err
python tools/ab_step.py --knob join --rounds 6 > $OUT/${TAG}_ab_join.json 2>> $OUT/bench.err
python tools/ab_step.py --knob bres --rounds 6 > $OUT/${TAG}_ab_bres.json 2>> $OUT/bench.err
python tools/ab_step.py --knob convstats --rounds 6 > $OUT/${TAG}_ab_convstats.json 2>> $OUT/bench.err
python tools/step_jitter.py 80 2>> $OUT/bench.err | tail -1 > $OUT/${TAG}_step_jitter.json
python tools/kernel_bench.py --cases a1,a3,a5,a2,a6,f4,pool,gemm16,dgrad16,fwdstats16,opt > $OUT/${TAG}_kernel_bench.jsonl 2> $OUT/kb.err
python tools/config4_bench.py > $OUT/${TAG}_config4_line.json 2> $OUT/c4.err
PMC_STEPS=3 bash tools/pmc_run.sh bench python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-recall --no-alone --no-mfu --no-prewarm > /dev/null 2>&1
cp $ROOT/gpurun_out/pmc_bench/summary.json $OUT/${TAG}_pmc_bench_traffic.json
bash tools/pmc_run.sh a3 python $ROOT/tools/kernel_bench.py --cases a3one > /dev/null 2>&1
cp $ROOT/gpurun_out/pmc_a3/summary.json $OUT/${TAG}_pmc_a3.json
bash tools/pmc_sq.sh a3 python $ROOT/tools/kernel_bench.py --cases a3one > $OUT/${TAG}_sq_a3.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace_a3 $OUT/trace_bench $OUT/trace_c4
cat > $OUT/pick.py <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'cfl_' in r['Name']]
w = csv.writer(sys.stdout)
w.writerow(['Name', 'Calls', 'AverageNs', 'MinNs', 'MaxNs'])
for r in rows:
    w.writerow([r['Name'].split('(float')[0].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', ''), r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs']])
PY
rocprofv3 --kernel-trace --stats -d $OUT/trace_a3one -o a3 --output-format csv -- python $ROOT/tools/kernel_bench.py --cases a3one > $OUT/trace_a3one.log 2>&1
python3 $OUT/pick.py $OUT/trace_a3one/a3_kernel_stats.csv > $OUT/${TAG}_a3one_kernel_stats.csv
CFL_BANK_NOIMG=1 rocprofv3 --kernel-trace --stats -d $OUT/trace_a3old -o a3 --output-format csv -- python $ROOT/tools/kernel_bench.py --cases a3one > $OUT/trace_a3old.log 2>&1
python3 $OUT/pick.py $OUT/trace_a3old/a3_kernel_stats.csv > $OUT/${TAG}_a3one_round2_path_kernel_stats.csv
rm -rf $OUT/trace_a3one $OUT/trace_a3old
rocprofv3 --kernel-trace --stats -d $OUT/trace_a3 -o a3 --output-format csv -- python $ROOT/tools/kernel_bench.py --cases a3 > $OUT/trace_a3.log 2>&1
python3 - $OUT/trace_a3/a3_kernel_stats.csv > $OUT/${TAG}_a3_kernel_stats.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'cfl_' in r['Name']]
w = csv.writer(sys.stdout)
w.writerow(['Name', 'Calls', 'AverageNs', 'MinNs', 'MaxNs'])
for r in rows:
    w.writerow([r['Name'].split('(float')[0].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', ''), r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs']])
PY
rocprofv3 --kernel-trace -d $OUT/trace_bench -o bench --output-format csv -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-recall --no-alone --no-prewarm > $OUT/trace_bench.log 2>&1
python3 $ROOT/tools/trace_stats.py $(ls $OUT/trace_bench/*kernel_trace.csv $OUT/trace_bench/*/*kernel_trace.csv 2>/dev/null | head -1) > $OUT/${TAG}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $OUT/trace_c4 -o c4 --output-format csv -- python $ROOT/tools/config4_bench.py --steps 5 --warmup 2 > $OUT/trace_c4.log 2>&1
python3 - $OUT/trace_c4/c4_kernel_stats.csv > $OUT/${TAG}_config4_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))[:25]
w = csv.writer(sys.stdout)
w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage'])
for r in rows:
    w.writerow([r['Name'][:110], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage']])
PY
rm -rf $OUT/trace_a3 $OUT/trace_bench $OUT/trace_c4
# last, bounded twice (its own watchdog + timeout): the self-launched 2-rank smoke run of the multi-GPU path on this one GPU
cd $ROOT
timeout 420 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-recall --watchdog 240 > $OUT/${TAG}_bench_gloo2_line.json 2> $OUT/gloo2.err
ls -la $OUT
