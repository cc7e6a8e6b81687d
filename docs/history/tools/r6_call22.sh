#!/bin/bash
# round 6, call 22: the stride-2 forward of the clients' 3x3 convolutions on the x3 kernel: parity, client steps with / without
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call22; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
export MIOPEN_LOG_LEVEL=1
( time timeout 900 python -m pytest tests/test_gpu_x3conv.py -q ) > $OUT/x3_tests.log 2>&1; tail -n 6 $OUT/x3_tests.log
( time timeout 1500 python -m pytest tests/test_gpu_framework.py -q -k "layout or client or x3" ) > $OUT/fw_tests.log 2>&1; tail -n 4 $OUT/fw_tests.log
for v in s2 lib s2 lib; do
  if [ $v = lib ]; then export CFL_NO_X3CONV_S2=1; else unset CFL_NO_X3CONV_S2; fi
  timeout 900 python bench.py --config 2 --round none --steps 30 --warmup 5 --no-cpu-baseline --only-kinds img,mm > $OUT/c2_$v.json 2> $OUT/c2_$v.err
  python3 -c "
import json
d=json.load(open('$OUT/c2_$v.json'))
print('$v', {k:(v.get('graph') or v.get('eager') or {}).get('ms_per_step') for k,v in d['clients'].items()})"
done
