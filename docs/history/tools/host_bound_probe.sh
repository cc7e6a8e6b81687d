#!/bin/bash
# How far is the server step from host-bound?  The bench step with this process confined to fewer and fewer
# host cycles: all cores / 2 cores / ONE core (main thread and autograd's backward thread share it) / one core shared with a busy
# competitor (= half a core per rank: 8 ranks on 4 free cores).  One JSON line per setting (ms per step of the same 12 steps).
#   bash tools/host_bound_probe.sh > profiles/r5_host_bound.jsonl
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
ARGS="--steps 12 --warmup 4 --no-cpu-baseline --no-recall --no-alone --no-mfu"
run() {   # label, prefix...
  label=$1; shift
  line=$("$@" python bench.py $ARGS 2>/dev/null | tail -1)
  python3 - "$label" "$line" <<'PY'
import json, sys
label, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    print(json.dumps({'setting': label, 'ms_per_step': d['ms_per_step'], 'pairs_per_s': d['value']}))
except Exception as e:
    print(json.dumps({'setting': label, 'error': str(e)[:100]}))
PY
}
run "all cores" env
run "2 cores (taskset -c 0,1)" taskset -c 0,1
run "1 core (taskset -c 0)" taskset -c 0
# a busy competitor on the same core: the rank gets about half of it
taskset -c 0 python3 -c "
import time
t=time.time()
while time.time()-t < 170: pass
" &
BUSY=$!
run "1 core shared with a busy process (half a core)" taskset -c 0
kill $BUSY 2>/dev/null
wait $BUSY 2>/dev/null
