#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_third
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/bnwg_probe tools/hip/bnwg_probe.hip 2> $OUT/probe_build.err
timeout 300 /tmp/bnwg_probe > $OUT/bnwg_probe.jsonl 2> $OUT/bnwg_probe.err
cat $OUT/bnwg_probe.jsonl
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_bnorm.py > $OUT/bnorm_tests.log 2>&1; tail -3 $OUT/bnorm_tests.log
( time timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_multirank.py ) > $OUT/multirank.log 2>&1; tail -12 $OUT/multirank.log
timeout 900 python tools/train_outcome_probe.py --steps 150 --batch 64 --n-id 1000 --lr 1e-3 > $OUT/outcome_a.jsonl 2> $OUT/outcome.err
cat $OUT/outcome_a.jsonl | cut -c1-400
timeout 900 python tools/train_outcome_probe.py --steps 300 --batch 128 --n-id 1000 --lr 1e-3 > $OUT/outcome_b.jsonl 2>> $OUT/outcome.err
cat $OUT/outcome_b.jsonl | cut -c1-400
tail -3 $OUT/outcome.err
( time timeout 1800 python bench.py --config 2 --steps 30 --warmup 5 ) > $OUT/config2_line.json 2> $OUT/config2.err
tail -5 $OUT/config2.err
cat $OUT/config2_line.json | cut -c1-3000
timeout 900 bash tools/host_bound_probe.sh > $OUT/host_bound.jsonl 2> $OUT/host_bound.err
cat $OUT/host_bound.jsonl
