#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5b5
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
run() { name=$1; shift; ( timeout 420 "$@" ) > $OUT/$name.log 2>&1; echo "== $name rc=$?"; grep -E "passed|failed|^E  |Fatal|Error" $OUT/$name.log | head -8 | cut -c1-400; }
run mm_graph python -m pytest tests/test_gpu_framework.py -m gpu -q -x -k "mm_client_contrast_step_in_a_hip_graph"
run server_graph python -m pytest tests/test_gpu_framework.py -m gpu -q -k "server_contrastive_step_in_a_hip_graph or dropout_masks"
run round_graph python -m pytest tests/test_gpu_framework.py -m gpu -q -x -k "one_communication_round"
timeout 420 python bench.py --config 2 --steps 30 --warmup 5 --no-cpu-baseline --server-graph 1 > $OUT/r5_config2_graphs_line.json 2> $OUT/c2g.err
python3 - <<'PY'
import json
for f in ('r5_config2_graphs_line.json',):
    try:
        d = json.loads(open('gpurun_out/r5b5/' + f).read().strip().splitlines()[-1])
        print(f, {k: (v['eager']['ms_per_step'], v.get('graph') and (v['graph']['ms_per_step'], v['graph']['capture_failed'])) for k, v in d['clients'].items()},
              d['round']['phases_s_rank0'], d['round']['ms_per_public_batch'], d['round'].get('graphs'))
    except Exception as e:
        print(f, 'FAILED', repr(e)[:200])
PY
tail -5 $OUT/c2g.err | cut -c1-300
