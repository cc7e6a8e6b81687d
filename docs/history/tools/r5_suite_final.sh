#!/bin/bash
# the round's last lines: bench first (fresh box), config 2 at the defaults, then the whole GPU suite + smoke
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_final
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/r5_bench_line_last.json 2> $OUT/bench.err
timeout 600 python bench.py --config 2 --steps 30 --warmup 5 > $OUT/r5_config2_line_last.json 2> $OUT/c2.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_final/r5_bench_line_last.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['alone']['frac'], d['mfu']['mfu'], d['cpu_baseline']['value'])
d=json.loads(open('gpurun_out/r5_final/r5_config2_line_last.json').read().strip().splitlines()[-1])
print('config2', d['value'], {k:(v['eager']['ms_per_step'], v.get('graph') and v['graph']['ms_per_step']) for k,v in d['clients'].items()}, d['round']['phases_s_rank0'], d['round']['ms_per_public_batch'], d['round'].get('graphs'), d['roofline']['frac'], d['cpu_baseline'] and d['cpu_baseline']['value'])
PY
( time python -m pytest tests -m gpu -q --durations=6 ) > $OUT/gputest.log 2>&1
tail -14 $OUT/gputest.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
