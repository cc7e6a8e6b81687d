#!/bin/bash
# round 6, final code: what the driver runs at round end (GPU suite, smoke, bench with its default command) + the config-2 line,
# then the kernel trace and the PMC traffic of the bench step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/final; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
export MIOPEN_LOG_LEVEL=1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r6_bench_line_driver.json 2> $OUT/bench.err
python3 -c "
import json
d=json.load(open('$OUT/r6_bench_line_driver.json')); r=d['roofline']
print('bench', d['ms_per_step'], d['value'], r['frac'], r['avg_launch_us'], r['traffic'], d['cpu_baseline']['value'], {k:(v.get('ms_per_step') if isinstance(v,dict) else None) for k,v in d['extra']['client_steps'].items() if k in ('image','text','multi_modal')}); print(d['extra'].get('hot_kernels'))"
tail -n 2 $OUT/bench.err
( time timeout 1500 python -m pytest tests -q -m gpu ) > $OUT/r6_gputest.log 2>&1
tail -n 6 $OUT/r6_gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -n 2 $OUT/smoke.log
timeout 1500 python bench.py --config 2 --no-cpu-baseline > $OUT/r6_config2_line.json 2> $OUT/config2.err
python3 -c "
import json
d=json.load(open('$OUT/r6_config2_line.json'))
print({k:(v.get('graph') or v.get('eager') or {}).get('ms_per_step') for k,v in d['clients'].items()})
print(d['round']['ms_per_public_batch'], d['round']['phases_s_rank0'])"
tail -n 3 $OUT/config2.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace_bench -o bench --output-format csv -- python3 $ROOT/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-client-steps --no-hot-kernels --no-recall --no-mfu --no-alone > $OUT/trace_bench.log 2>&1
ls $OUT/trace_bench | head
cd $ROOT
PMC_STEPS=7 bash tools/pmc_run.sh bench python3 $ROOT/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-client-steps --no-hot-kernels --no-recall --no-mfu --no-alone > $OUT/pmc_bench.log 2>&1
tail -n 12 $OUT/pmc_bench.log
timeout 900 python tools/config4_bench.py --batch 256 > $OUT/r6_config4_b256_line.json 2> $OUT/c4.err; cat $OUT/r6_config4_b256_line.json
timeout 900 python tools/kernel_bench.py --cases a1,a3,a5,a5wide > $OUT/r6_kernel_bench_final.jsonl 2> $OUT/kb.err; cut -c1-330 $OUT/r6_kernel_bench_final.jsonl
