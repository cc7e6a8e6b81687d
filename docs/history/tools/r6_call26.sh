#!/bin/bash
# round 6, call 26: the 8-wave wide con_w kernel (D <= 256) without the per-burst liveness branch / with the issue order pinned
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call26; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
for v in n b p n b p; do
  CFL_CONW_WIDE_RB=$v timeout 600 python tools/kernel_bench.py --cases a5 2> $OUT/kb_$v.err | sed "s/^{/{\"form\": \"$v\", /" >> $OUT/r6_a5_wide32_branch_ab.jsonl
done
cut -c1-330 $OUT/r6_a5_wide32_branch_ab.jsonl
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -k "a5 or conw or config2" ) > $OUT/conw_tests.log 2>&1; tail -n 4 $OUT/conw_tests.log
