#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call5; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
export MIOPEN_LOG_LEVEL=1
( time timeout 900 python -m pytest tests/test_gpu_x3conv.py -q ) > $OUT/x3_tests.log 2>&1; tail -n 4 $OUT/x3_tests.log
timeout 900 python tools/kernel_bench.py --cases x3conv,x3dbg > $OUT/x3conv_probe.jsonl 2> $OUT/x3conv_probe.err; tail -n 3 $OUT/x3conv_probe.err
python3 - <<'PY'
import json,os
for ln in open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/call5/x3conv_probe.jsonl'):
    if ln.startswith('{'):
        d=json.loads(ln); print(d['case'], {k:v for k,v in d.items() if k.endswith('fwd_us')})
PY
