#!/bin/bash
# round 6, call 9: con_w on the 4-wave wide bank kernel at 256 < D <= 512 (parity + timing vs the tile GEMM; burst lengths 2 / 4 / 8)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call9; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
( time timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "a5 or conw" ) > $OUT/conw_tests.log 2>&1; tail -n 6 $OUT/conw_tests.log
timeout 600 python tools/kernel_bench.py --cases a5,a5wide > $OUT/r6_a5_conw_lines.jsonl 2> $OUT/kb.err; tail -n 3 $OUT/kb.err
for rb in 2 8; do
  CFL_CONW_WIDE_RB=$rb timeout 600 python tools/kernel_bench.py --cases a5wide 2> $OUT/kb$rb.err | head -n 1 | sed "s/^{/{\"CFL_CONW_WIDE_RB\": $rb, /" >> $OUT/r6_a5_conw_lines.jsonl
done
cut -c1-420 $OUT/r6_a5_conw_lines.jsonl
