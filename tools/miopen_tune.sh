#!/bin/bash
# Exhaustive MIOpen tuning of every convolution problem of the bench step (MIOPEN_FIND_ENFORCE=SEARCH_DB_UPDATE): the user
# perf-db / find-db that comes out replaces creamfl_amd/miopen_db if the step gets faster with it.
#   bash tools/miopen_tune.sh          (via gpurun; results under gpurun_out/miopen_tuned)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
DB=$ROOT/gpurun_out/miopen_tuned
rm -rf $DB; mkdir -p $DB
cp $ROOT/creamfl_amd/miopen_db/* $DB/
cd $ROOT
export MIOPEN_USER_DB_PATH=$DB MIOPEN_LOG_LEVEL=1
echo "before (recorded db):"; python bench.py --steps 20 --warmup 5 --no-recall --no-cpu-baseline --no-alone 2>/dev/null | tail -1 | cut -c60-200
t0=$(date +%s)
MIOPEN_FIND_MODE=1 MIOPEN_FIND_ENFORCE=4 timeout ${TUNE_TIMEOUT:-2400} python bench.py --steps 2 --warmup 1 --no-recall --no-cpu-baseline --no-alone > $DB/tune.log 2>&1
echo "tuning took $(( $(date +%s) - t0 )) s, rc=$?"
ls -la $DB
echo "after (tuned db):"; python bench.py --steps 20 --warmup 5 --no-recall --no-cpu-baseline --no-alone 2>/dev/null | tail -1 | cut -c60-200
python bench.py --steps 20 --warmup 5 --no-recall --no-cpu-baseline --no-alone 2>/dev/null | tail -1 | cut -c60-200
