#!/bin/bash
# round 6, call 14: the bank stream kernel with its LDS-DMA issued from inline assembly (no compiler vmcnt(0) in front of the transposing
# reads): parity (every A3 / A4 / A5 test), kernel bench A/B against the builtin form; ViT chain tests after the fixes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call14; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
export MIOPEN_LOG_LEVEL=1
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -q -k "a3 or a34 or a5 or conw or contrast or bank" ) > $OUT/bank_tests.log 2>&1; tail -n 6 $OUT/bank_tests.log
for v in asm builtin asm builtin; do
  if [ $v = builtin ]; then export CFL_BANK_DMA_BUILTIN=1; else unset CFL_BANK_DMA_BUILTIN; fi
  timeout 600 python tools/kernel_bench.py --cases a3 2> $OUT/kb_$v.err | sed "s/^{/{\"dma\": \"$v\", /" >> $OUT/r6_a3_dma_ab.jsonl
done
unset CFL_BANK_DMA_BUILTIN
python3 - <<P
import json
for l in open('$OUT/r6_a3_dma_ab.jsonl'):
    d=json.loads(l); print(d.get('dma'), d.get('case'), d.get('kernels_us'), d.get('us_per_step'))
P
( time timeout 900 python -m pytest tests/test_gpu_bert.py -q -k "preln or vit" ) > $OUT/vit_tests.log 2>&1; tail -n 6 $OUT/vit_tests.log
( time timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_framework.py -q -k "config4" ) > $OUT/c4_tests.log 2>&1; tail -n 4 $OUT/c4_tests.log
