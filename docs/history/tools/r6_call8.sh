#!/bin/bash
# round 6, call 8: version 4 of the x3 forward kernel (loader roles per wave) — tests + kernel bench at the four BasicBlock shapes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call8; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
export MIOPEN_LOG_LEVEL=1
( time timeout 900 python -m pytest tests/test_gpu_x3conv.py -q ) > $OUT/x3_tests.log 2>&1; tail -n 6 $OUT/x3_tests.log
timeout 900 python tools/kernel_bench.py --cases x3conv > $OUT/r6_x3conv_probe_v4.jsonl 2> $OUT/kb.err; tail -n 3 $OUT/kb.err
python3 - <<P
import json
for l in open('$OUT/r6_x3conv_probe_v4.jsonl'):
    d=json.loads(l); print(d['case'], {k:v for k,v in d.items() if k.endswith('fwd_us') or k.endswith('dgrad_us')})
P
