#!/bin/bash
# GPU call 1 of the round's last session: the new graph paths (AdamP in a graph, multi-modal client, server steps), their A/B in
# a config 2 round, con_w at D = 512 / 768
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5b
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
( time timeout 600 python -m pytest tests/test_gpu_optimizer.py tests/test_gpu_framework.py -m gpu -q -x \
    -k "graph or dropout or adamp or daln or bert" --durations=5 ) > $OUT/tests_new.log 2>&1
tail -25 $OUT/tests_new.log | cut -c1-400
timeout 200 python tools/kernel_bench.py --cases a5wide > $OUT/r5_a5_conw_wide.jsonl 2> $OUT/a5.err; tail -3 $OUT/r5_a5_conw_wide.jsonl | cut -c1-400
timeout 420 python bench.py --config 2 --steps 30 --warmup 5 --no-cpu-baseline --server-graph 1 > $OUT/r5_config2_graphs_line.json 2> $OUT/c2g.err
timeout 420 python bench.py --config 2 --steps 30 --warmup 5 --no-cpu-baseline --server-graph 0 --mm-client-graph 0 > $OUT/r5_config2_nograph_line.json 2> $OUT/c2n.err
python3 - <<'PY'
import json
for f in ('r5_config2_graphs_line.json', 'r5_config2_nograph_line.json'):
    try:
        d = json.loads(open('gpurun_out/r5b/' + f).read().strip().splitlines()[-1])
        print(f, {k: (v['eager']['ms_per_step'], v.get('graph') and (v['graph']['ms_per_step'], v['graph']['capture_failed'])) for k, v in d['clients'].items()},
              d['round']['phases_s_rank0'], d['round']['ms_per_public_batch'], d['round'].get('graphs'))
    except Exception as e:
        print(f, 'FAILED', repr(e)[:200])
PY
tail -5 $OUT/c2g.err | cut -c1-300
