#!/bin/bash
# Record MIOpen's find-db / perf-db / compiled-kernel cache for the convolution problems of a configs[2] round (server at batch
# 128 / 256, the three client kinds at 128 / 256 / 512, 224 x 224) on this box: every problem goes through the timed search
# (CFL_MIOPEN_AUTO=0) into a FRESH user directory that starts from the shipped files.  tools/merge_find_db.py then adds the NEW
# records to creamfl_amd/miopen_db / miopen_cache (records the package already ships are kept verbatim: they were measured).
#   gpurun -- 'bash tools/record_find_db.sh'   ->  gpurun_out/fdb_record/{db,cache}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fdb_record
rm -rf $OUT; mkdir -p $OUT/db $OUT/cache
cp $ROOT/creamfl_amd/miopen_db/* $OUT/db/
cp $ROOT/creamfl_amd/miopen_cache/* $OUT/cache/
cd $ROOT
export CFL_NO_SEEDED_DB=1 CFL_MIOPEN_AUTO=0 MIOPEN_USER_DB_PATH=$OUT/db MIOPEN_CUSTOM_CACHE_DIR=$OUT/cache MIOPEN_LOG_LEVEL=1
( time timeout 2400 python bench.py --config 2 --steps 3 --warmup 2 --round full --round-pub 512 --no-cpu-baseline ) > $OUT/line.json 2> $OUT/err.log
tail -3 $OUT/err.log
ls -la $OUT/db $OUT/cache
wc -l $OUT/db/*.txt
rm -f $OUT/cache/*-wal $OUT/cache/*-shm
