#!/bin/bash
# Round-6 step accounting in one gpurun call:  bash tools/r6_account.sh  -> gpurun_out/r6_account/
#   r6_bench_line_first.json   python bench.py --steps 20 --warmup 5 (this lease's baseline)
#   r6_bound_{wgrad,text,sides}.json   tools/ab_step.py bounds: the step with the side streams' work replaced by nothing
#   r6_step_bytes.json         PMC bytes of EVERY kernel of the step (library included), per queue / family, + whole-step floors
#   r6_bench_streams.json      per-queue busy time and idle gaps from a kernel trace
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_account
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r6_bench_line_first.json 2> $OUT/bench.err
for K in wgrad text sides; do
  python tools/ab_step.py --knob $K --rounds 4 > $OUT/r6_bound_$K.json 2>> $OUT/ab.err
done
PMC_STEPS=3 bash tools/pmc_run.sh bench python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-recall --no-alone --no-mfu > /dev/null 2>&1
cp $ROOT/gpurun_out/pmc_bench/summary.json $OUT/r6_pmc_bench_traffic.json
STEP_MS=$(python3 -c "import json;print(json.loads(open('$OUT/r6_bench_line_first.json').read().strip().splitlines()[-1])['ms_per_step'])")
python3 tools/step_bytes.py $ROOT/gpurun_out/pmc_bench --step-ms $STEP_MS --tflop 15.08 > $OUT/r6_step_bytes.json 2> $OUT/step_bytes.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace_bench -o bench --output-format csv -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-recall --no-alone --no-mfu > $OUT/trace_bench.log 2>&1
T=$(ls $OUT/trace_bench/*kernel_trace.csv $OUT/trace_bench/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 $ROOT/tools/trace_streams.py $T > $OUT/r6_bench_streams.json
python3 $ROOT/tools/trace_stats.py $T > $OUT/r6_bench_kernel_stats.csv
rm -rf $OUT/trace_bench $ROOT/gpurun_out/pmc_bench
ls -la $OUT; cat $OUT/r6_bound_*.json; head -40 $OUT/r6_step_bytes.json
