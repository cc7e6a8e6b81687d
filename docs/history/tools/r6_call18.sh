#!/bin/bash
# round 6, call 18: con_w at D <= 256 with 64 rows of V per wave (cfl_bank_wide64_kernel): parity + timing vs the 8-wave form
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call18; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -k "a5 or conw or config2" ) > $OUT/conw_tests.log 2>&1; tail -n 6 $OUT/conw_tests.log
for v in w64 w32 w64 w32; do
  if [ $v = w32 ]; then export CFL_CONW_WIDE64=0; else unset CFL_CONW_WIDE64; fi
  timeout 600 python tools/kernel_bench.py --cases a5 2> $OUT/kb_$v.err | sed "s/^{/{\"form\": \"$v\", /" >> $OUT/r6_a5_wide64_ab.jsonl
done
unset CFL_CONW_WIDE64
cut -c1-420 $OUT/r6_a5_wide64_ab.jsonl
