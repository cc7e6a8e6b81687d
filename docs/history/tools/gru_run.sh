set -x
mkdir -p gpurun_out/bh
python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bh/c2_freeze.json 2> gpurun_out/bh/c2_freeze.err
CFL_NO_GC_FREEZE=1 python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bh/c2_nofreeze.json 2> gpurun_out/bh/c2_nofreeze.err
python - <<'PY'
import json
for f in ('c2_freeze','c2_nofreeze'):
    d=json.loads(open('gpurun_out/bh/%s.json'%f).read().strip().splitlines()[-1])
    print(f, d['round']['phases_s_rank0'], d['round']['ms_per_public_batch']); print('   first', d['round']['first_round_phases_s_rank0'])
PY
