#!/bin/bash
# Record the MIOpen find-db / perf-db entries of every convolution problem of the bench step that the shipped
# creamfl_amd/miopen_db does not have yet (normal find, no forced re-tuning).  Result: gpurun_out/miopen_db_new/ -- copy the two
# .txt files over creamfl_amd/miopen_db/ if the step is faster with them.     bash tools/miopen_record.sh   (via gpurun)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
DB=$ROOT/gpurun_out/miopen_db_new
rm -rf $DB; mkdir -p $DB
cp $ROOT/creamfl_amd/miopen_db/* $DB/
cd $ROOT
export MIOPEN_USER_DB_PATH=$DB MIOPEN_LOG_LEVEL=1
echo "shipped db:"; python bench.py --steps 20 --warmup 5 --no-recall --no-cpu-baseline --no-alone 2>/dev/null | tail -1 | cut -c60-200
MIOPEN_FIND_MODE=1 python bench.py --steps 2 --warmup 1 --no-recall --no-cpu-baseline --no-alone > $DB/record.log 2>&1
wc -l $DB/*.txt
echo "recorded db:"; python bench.py --steps 20 --warmup 5 --no-recall --no-cpu-baseline --no-alone 2>/dev/null | tail -1 | cut -c60-200
python bench.py --steps 20 --warmup 5 --no-recall --no-cpu-baseline --no-alone 2>/dev/null | tail -1 | cut -c60-200
echo "recorded db, stem as the library sees it:"; CFL_NO_STEM_S2D=1 python bench.py --steps 20 --warmup 5 --no-recall --no-cpu-baseline --no-alone 2>/dev/null | tail -1 | cut -c60-200
