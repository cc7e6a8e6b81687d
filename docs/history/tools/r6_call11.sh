#!/bin/bash
# round 6, call 11: con_w at D = 768 on the 16-row-step wide kernel (parity + timing vs the tile GEMM)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call11; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
( time timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "a5 or conw" ) > $OUT/conw_tests.log 2>&1; tail -n 8 $OUT/conw_tests.log
timeout 600 python tools/kernel_bench.py --cases a5wide > $OUT/r6_a5_conw_768.jsonl 2> $OUT/kb.err; tail -n 3 $OUT/kb.err
cut -c1-420 $OUT/r6_a5_conw_768.jsonl
