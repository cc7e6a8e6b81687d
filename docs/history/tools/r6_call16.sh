#!/bin/bash
# round 6, call 16: A1 backward on 256 x 256 tiles + split-K (image mode, N % 256 == 0, D % 256 == 0): parity + kernel bench A/B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call16; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -k "a1 or config3 or pair" ) > $OUT/a1_tests.log 2>&1; tail -n 8 $OUT/a1_tests.log
for v in big ring big ring; do
  if [ $v = ring ]; then export CFL_PAIR_BWD_BIG=0; else unset CFL_PAIR_BWD_BIG; fi
  timeout 600 python tools/kernel_bench.py --cases a1 2> $OUT/kb_$v.err | tail -n 1 | sed "s/^{/{\"bwd\": \"$v\", /" >> $OUT/r6_a1_bwd_ab.jsonl
done
unset CFL_PAIR_BWD_BIG
cut -c1-400 $OUT/r6_a1_bwd_ab.jsonl
