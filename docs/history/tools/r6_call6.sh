#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call6; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
export MIOPEN_LOG_LEVEL=1
( time timeout 900 python -m pytest tests/test_gpu_bnorm.py -q -k "stem_tail" ) > $OUT/stem_tests.log 2>&1; tail -n 12 $OUT/stem_tests.log
( time timeout 900 python -m pytest tests/test_gpu_x3conv.py -q ) > $OUT/x3_tests.log 2>&1; tail -n 4 $OUT/x3_tests.log
( time timeout 1200 python -m pytest tests/test_gpu_framework.py tests/test_gpu_parity.py -q -k "layout or client or a2c or tower" ) > $OUT/fw_tests.log 2>&1; tail -n 8 $OUT/fw_tests.log
timeout 1500 python bench.py --config 2 --no-cpu-baseline --steps 10 --warmup 3 --round none > $OUT/config2_line.json 2> $OUT/config2.err
python3 -c "
import json
d=json.load(open('$OUT/config2_line.json'))
print({k:(v.get('graph') or v.get('eager') or {}).get('ms_per_step') for k,v in d['clients'].items()})
print({k:v['us_per_step'] for k,v in d['clients']['img']['hip_kernels'].items()})"
tail -n 3 $OUT/config2.err
