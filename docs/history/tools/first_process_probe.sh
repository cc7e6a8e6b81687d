#!/bin/bash
# Does the library set-up of creamfl_amd/runtime.py (recorded find-db + shipped kernel cache) make the FIRST process on a fresh
# box run the step as fast as a later one?  Three driver-style invocations of bench.py without the pre-warm child:
#   A  fresh box, shipped find-db + shipped kernel cache (the product default)
#   B  the same again (second process)
#   C  everything the library wrote wiped, shipped find-db but an EMPTY kernel cache directory (what round 3 had)
# usage (gpurun): bash tools/first_process_probe.sh > gpurun_out/first_process.txt
export MIOPEN_LOG_LEVEL=1
run() {
  local t0=$SECONDS
  "$@" python bench.py --no-prewarm --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-alone 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('ms_per_step', d['ms_per_step'], 'pairs/s', d['value'])"
  echo "wall_s $((SECONDS - t0))"
}
echo "== A first process, shipped find-db + kernel cache"; run env
echo "== B second process"; run env
rm -rf ~/.cache/miopen ~/.cache/comgr /tmp/creamfl_miopen_db_* /tmp/creamfl_miopen_cache_* /tmp/emptycache; mkdir -p /tmp/emptycache
echo "== C first process again (library caches wiped), shipped find-db, EMPTY kernel cache"; run env MIOPEN_CUSTOM_CACHE_DIR=/tmp/emptycache
echo "== D second process of C"; run env MIOPEN_CUSTOM_CACHE_DIR=/tmp/emptycache
