#!/bin/bash
# round 6, call 17: the bank stream kernel at D > 256 with one wave per SIMD (no column-split pairs): parity + kernel bench A/B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call17; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
export MIOPEN_LOG_LEVEL=1
( time CFL_BANK_1WAVE=1 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -k "a3 or a34 or contrast or bank or config4" ) > $OUT/bank_tests.log 2>&1; tail -n 6 $OUT/bank_tests.log
for v in one pair one pair; do
  if [ $v = one ]; then export CFL_BANK_1WAVE=1; else unset CFL_BANK_1WAVE; fi
  timeout 600 python tools/kernel_bench.py --cases a3 2> $OUT/kb_$v.err | grep -E "D=512|D=768" | sed "s/^{/{\"waves\": \"$v\", /" >> $OUT/r6_a3_1wave_ab.jsonl
done
unset CFL_BANK_1WAVE
python3 - <<P
import json
for l in open('$OUT/r6_a3_1wave_ab.jsonl'):
    d=json.loads(l); print(d.get('waves'), d.get('case'), d.get('kernels_us'), d.get('us_per_step'))
P
