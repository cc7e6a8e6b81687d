#!/bin/bash
# round 6, call 20: A1 forward with the lean per-pair arithmetic on interior off-diagonal tiles: parity + kernel bench
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call20; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_framework.py -q -k "a1 or config3 or pair or s1 or stage" ) > $OUT/a1_tests.log 2>&1; tail -n 8 $OUT/a1_tests.log
timeout 600 python tools/kernel_bench.py --cases a1 > $OUT/r6_a1_lean.jsonl 2> $OUT/kb.err; cut -c1-400 $OUT/r6_a1_lean.jsonl
