#!/bin/bash
# round 6, call 25: evaluation-mode BatchNorm in the epilogue of the x3 convolutions (old model / representation extraction): parity, client steps A/B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call25; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
export MIOPEN_LOG_LEVEL=1
( time timeout 900 python -m pytest tests/test_gpu_x3conv.py -q ) > $OUT/x3_tests.log 2>&1; tail -n 12 $OUT/x3_tests.log
( time timeout 1500 python -m pytest tests/test_gpu_framework.py tests/test_gpu_parity.py -q -k "layout or client or x3 or a2c or mirror" ) > $OUT/fw_tests.log 2>&1; tail -n 4 $OUT/fw_tests.log
for v in epi two epi two; do
  if [ $v = two ]; then export CFL_NO_X3_BN_EPI=1; else unset CFL_NO_X3_BN_EPI; fi
  timeout 900 python bench.py --config 2 --round none --steps 30 --warmup 5 --no-cpu-baseline --only-kinds img,mm > $OUT/c2_$v.json 2> $OUT/c2_$v.err
  python3 -c "
import json
d=json.load(open('$OUT/c2_$v.json'))
print('$v', {k:((v.get('graph') or {}).get('ms_per_step'), v['eager']['ms_per_step']) for k,v in d['clients'].items()})"
done
