#!/bin/bash
# Round-5 evidence in one gpurun call (run from the repo root):  bash tools/profile_r5.sh [tag]   -> gpurun_out/profiles_<tag>/
#   <tag>_bench_line.json             python bench.py --steps 20 --warmup 5 (the driver's command line)
#   <tag>_bench_line_again.json       the same, GPU part only (second process on the box)
#   <tag>_ab_bnslice.json             tools/ab_step.py --knob bnslice: channel-sliced BatchNorm map on / off, same process
#   <tag>_config2_line.json           python bench.py --config 2: client contrast steps + one round + the full-M exchange
#   <tag>_client_step_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the client contrast steps (config 2, no round)
#   <tag>_bench_kernel_stats.csv      rocprofv3 --kernel-trace of a short bench run, timed steps only (tools/trace_stats.py)
#   <tag>_pmc_bench_traffic.json      HBM traffic per hand-written kernel of the bench step (separate --pmc passes)
#   <tag>_kernel_bench.jsonl          tools/kernel_bench.py --cases a3,a5 (HIP-event numbers at the SURVEY 8(d) shapes)
#   <tag>_a3one_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the client contrast kernels, B = 128, M = 50 000, D = 256
#   <tag>_pmc_a3.json, <tag>_sq_a3.json   HBM traffic and SQ / LDS / MFMA counters of the same kernels at the final code
#   <tag>_host_bound.jsonl            tools/host_bound_probe.sh
#   <tag>_train_outcome.jsonl         tools/train_outcome_probe.py (bf16 fused vs fp32 trunks on the learnable task)
TAG=${1:-r5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall > $OUT/${TAG}_bench_line_again.json 2>> $OUT/bench.err
python tools/ab_step.py --knob bnslice --rounds 6 > $OUT/${TAG}_ab_bnslice.json 2>> $OUT/bench.err
timeout 1800 python bench.py --config 2 --steps 30 --warmup 5 > $OUT/${TAG}_config2_line.json 2> $OUT/config2.err
python tools/kernel_bench.py --cases a3,a5 > $OUT/${TAG}_kernel_bench.jsonl 2> $OUT/kb.err
python tools/train_outcome_probe.py > $OUT/${TAG}_train_outcome.jsonl 2> $OUT/outcome.err
bash tools/host_bound_probe.sh > $OUT/${TAG}_host_bound.jsonl 2> $OUT/host.err
PMC_STEPS=3 bash tools/pmc_run.sh bench python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-recall --no-alone --no-mfu > /dev/null 2>&1
cp $ROOT/gpurun_out/pmc_bench/summary.json $OUT/${TAG}_pmc_bench_traffic.json
bash tools/pmc_run.sh a3 python $ROOT/tools/kernel_bench.py --cases a3one > /dev/null 2>&1
cp $ROOT/gpurun_out/pmc_a3/summary.json $OUT/${TAG}_pmc_a3.json
bash tools/pmc_sq.sh a3 python $ROOT/tools/kernel_bench.py --cases a3one > $OUT/${TAG}_sq_a3.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
cat > $OUT/pick.py <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if len(sys.argv) > 2 or 'cfl_' in r['Name']]
w = csv.writer(sys.stdout)
w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
for r in rows:
    w.writerow([r['Name'].split('(float')[0].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:140], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
PY
rocprofv3 --kernel-trace --stats -d $OUT/trace_a3one -o a3 --output-format csv -- python $ROOT/tools/kernel_bench.py --cases a3one > $OUT/trace_a3one.log 2>&1
python3 $OUT/pick.py $(ls $OUT/trace_a3one/*kernel_stats.csv $OUT/trace_a3one/*/*kernel_stats.csv 2>/dev/null | head -1) > $OUT/${TAG}_a3one_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $OUT/trace_client -o cl --output-format csv -- python $ROOT/bench.py --config 2 --round none --steps 20 --warmup 3 --no-cpu-baseline > $OUT/trace_client.log 2>&1
python3 $OUT/pick.py $(ls $OUT/trace_client/*kernel_stats.csv $OUT/trace_client/*/*kernel_stats.csv 2>/dev/null | head -1) all > $OUT/${TAG}_client_step_kernel_stats.csv
rocprofv3 --kernel-trace -d $OUT/trace_bench -o bench --output-format csv -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-recall --no-alone > $OUT/trace_bench.log 2>&1
python3 $ROOT/tools/trace_stats.py $(ls $OUT/trace_bench/*kernel_trace.csv $OUT/trace_bench/*/*kernel_trace.csv 2>/dev/null | head -1) > $OUT/${TAG}_bench_kernel_stats.csv
rm -rf $OUT/trace_a3one $OUT/trace_client $OUT/trace_bench
ls -la $OUT
cd $ROOT
( time python -m pytest tests/test_gpu_multirank.py -m gpu -x -q ) > $OUT/multirank_tests.log 2>&1
tail -5 $OUT/multirank_tests.log
