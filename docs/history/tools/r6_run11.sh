#!/bin/bash
# round 6, GPU call 11: slab version of the 3 x bf16-split convolution: parity, probe, client steps
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_run11
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
( timeout 900 python -m pytest tests/test_gpu_x3conv.py -q -m gpu ) > $OUT/test_x3.log 2>&1
tail -n 12 $OUT/test_x3.log
timeout 1200 python tools/kernel_bench.py --cases x3conv > $OUT/r6_x3conv_probe.jsonl 2> $OUT/kb.err
python3 -c "
import json
for l in open('$OUT/r6_x3conv_probe.jsonl'):
    d=json.loads(l); print(d['case'], {k:v for k,v in d.items() if k.endswith('_us') or k.startswith('speedup') or k=='best_variant'})"
tail -n 3 $OUT/kb.err
timeout 1200 python bench.py --config 2 --round none --steps 30 --warmup 5 --no-cpu-baseline --client-conv-x3 1 > $OUT/c2_x3.json 2>> $OUT/c2.err
python3 -c "
import json
d=json.load(open('$OUT/c2_x3.json'))
print('x3', {k:((v.get('graph') or {}).get('ms_per_step'), (v.get('eager') or {}).get('ms_per_step')) for k,v in d['clients'].items()})"
tail -n 5 $OUT/c2.err
