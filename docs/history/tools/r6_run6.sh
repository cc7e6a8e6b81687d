#!/bin/bash
# round 6, GPU call 6: 1x1 weight-gradient knobs in the step (workgroup target, map-size gate), CU-masked side streams, the re-ordered
# config-2 bench, the three-seed outcome test
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_run6
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
for K in w1wgs256 w1wgs64 w1hw14 w1hw28; do
  timeout 600 python tools/ab_step.py --knob $K --rounds 4 > $OUT/r6_ab_$K.json 2>> $OUT/ab.err
  cat $OUT/r6_ab_$K.json
done
tail -n 3 $OUT/ab.err
B="python bench.py --no-cpu-baseline --no-client-steps --no-mfu --no-recall --steps 20 --warmup 8"
$B > $OUT/bench_base.json 2>> $OUT/bench.err
for N in 192 128 64; do CFL_WGRAD_CUS=$N $B > $OUT/bench_wcus$N.json 2>> $OUT/bench.err; done
CFL_TEXT_CUS=64 $B > $OUT/bench_tcus64.json 2>> $OUT/bench.err
CFL_TEXT_CUS=64 CFL_WGRAD_CUS=128 $B > $OUT/bench_tcus64_wcus128.json 2>> $OUT/bench.err
$B > $OUT/bench_base2.json 2>> $OUT/bench.err
for f in base wcus192 wcus128 wcus64 tcus64 tcus64_wcus128 base2; do python3 -c "
import json,sys
try:
    d=json.load(open('$OUT/bench_$f.json')); r=d.get('roofline') or {}
    print('$f', d['ms_per_step'], r.get('avg_launch_us'), r.get('frac'))
except Exception as e: print('$f', 'failed', e)"; done
tail -n 5 $OUT/bench.err
( timeout 900 python -m pytest tests/test_gpu_framework.py -q -m gpu -s -k "ambiguous" ) > $OUT/test_outcome3.log 2>&1
grep -E "training outcome|passed|failed|Error" $OUT/test_outcome3.log | cut -c1-600
timeout 1500 python bench.py --config 2 --no-cpu-baseline > $OUT/r6_config2_line.json 2> $OUT/config2.err
python3 -c "
import json
d=json.load(open('$OUT/r6_config2_line.json'))
print({k:(v.get('graph') or v.get('eager') or {}).get('ms_per_step') for k,v in d['clients'].items()})
print(d['round']['ms_per_public_batch'], d['round']['phases_s_rank0'])
print(d['roofline'])"
tail -n 3 $OUT/config2.err
# the server's global-training phase inside the federation process: producer thread vs inline loader, allocator knobs
T="timeout 600 python tools/federation_step_trace.py --batches 50 --rounds 2"
$T --tag base > $OUT/fed_base.json 2>> $OUT/fed.err
CFL_PREFETCH_INLINE=1 $T --tag inline > $OUT/fed_inline.json 2>> $OUT/fed.err
CFL_PREFETCH_INLINE=1 $T --tag inline_gcoff --gc-off 1 > $OUT/fed_inline_gcoff.json 2>> $OUT/fed.err
CFL_PREFETCH_INLINE=1 PYTORCH_HIP_ALLOC_CONF=expandable_segments:True PYTORCH_CUDA_ALLOC_CONF=expandable_segments:True $T --tag inline_expandable > $OUT/fed_inline_expandable.json 2>> $OUT/fed.err
cat $OUT/fed_*.json > $OUT/r6_federation_step_trace.jsonl
python3 -c "
import json
for l in open('$OUT/r6_federation_step_trace.jsonl'):
    d=json.loads(l)
    print(d['tag'], [(p['wall_ms_per_batch'], p['issue_ms_per_batch'], p['median_rest_ms'], p['p90_rest_ms'], p['mem_gb']['reserved']) for p in d['phases']])"
tail -n 3 $OUT/fed.err
# kernel trace of the bench step at this code: per-kernel statistics, per-queue busy time, the main queue's largest idle gaps
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace_bench -o bench --output-format csv -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-recall --no-alone --no-client-steps --no-mfu > $OUT/trace_bench.log 2>&1
TR=$(ls $OUT/trace_bench/*kernel_trace.csv $OUT/trace_bench/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 $ROOT/tools/trace_stats.py $TR > $OUT/r6_bench_kernel_stats.csv
python3 $ROOT/tools/trace_streams.py $TR --gaps 14 > $OUT/r6_bench_streams.json
rm -rf $OUT/trace_bench
head -n 12 $OUT/r6_bench_kernel_stats.csv | cut -c1-160
python3 -c "
import json
d=json.load(open('$OUT/r6_bench_streams.json'))
print(d['wall_ms_per_step'])
for q,v in d['queues'].items():
    print(q, v['busy_ms_per_step'], v['gaps']['idle_ms_per_step'])
    for g in v.get('largest_gaps', [])[:24]: print('   ', g)
" | head -80
