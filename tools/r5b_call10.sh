#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5b11
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
timeout 170 python bench.py --config 2 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/r5_config2_round_first_line.json 2> $OUT/c2.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5b11/r5_config2_round_first_line.json').read().strip().splitlines()[-1])
print('config2', d['value'], {k:(v['eager']['ms_per_step'], v.get('graph') and v['graph']['ms_per_step']) for k,v in d['clients'].items()}, d['round']['phases_s_rank0'], d['round']['ms_per_public_batch'], d['round'].get('graphs'), d['roofline']['frac'])
PY
tail -3 $OUT/c2.err | cut -c1-300
