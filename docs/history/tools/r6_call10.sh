#!/bin/bash
# round 6, call 10: issue-order variants of the 4-wave wide con_w kernel (sched_group_barrier: s = 3 MFMAs / 2 reads, t = 1 + 2 reads + 2)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call10; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
for rb in 4 s t; do
  CFL_CONW_WIDE_RB=$rb timeout 600 python tools/kernel_bench.py --cases a5wide 2> $OUT/kb$rb.err | head -n 1 | sed "s/^{/{\"CFL_CONW_WIDE_RB\": \"$rb\", /" >> $OUT/variants.jsonl
done
cut -c1-420 $OUT/variants.jsonl
CFL_CONW_WIDE_RB=s timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "a5 or conw" 2>&1 | tail -n 2
CFL_CONW_WIDE_RB=t timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "a5 or conw" 2>&1 | tail -n 2
