#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5b9
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
timeout 300 python tools/federation_step_trace.py 2> $OUT/t.err | tee $OUT/trace.jsonl | cut -c1-900
timeout 400 python bench.py --config 2 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/c2.json 2> $OUT/c2.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5b9/c2.json').read().strip().splitlines()[-1])
print('config2', d['round']['phases_s_rank0'], d['round']['ms_per_public_batch'], d['round']['first_round_phases_s_rank0'])
PY
