#!/bin/bash
# round 6, GPU call 10: the 3 x bf16-split convolution for the clients' fp32 encoders: parity, probe at the four BasicBlock shapes, the
# image client's step with it
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_run10
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
( timeout 900 python -m pytest tests/test_gpu_x3conv.py tests/test_gpu_wgrad.py -q -m gpu ) > $OUT/test_x3.log 2>&1
tail -n 12 $OUT/test_x3.log
timeout 1200 python tools/kernel_bench.py --cases x3conv > $OUT/r6_x3conv_probe.jsonl 2> $OUT/kb.err
cat $OUT/r6_x3conv_probe.jsonl; tail -n 3 $OUT/kb.err
timeout 1200 python bench.py --config 2 --round none --steps 30 --warmup 5 --no-cpu-baseline > $OUT/c2_lib.json 2> $OUT/c2.err
timeout 1200 python bench.py --config 2 --round none --steps 30 --warmup 5 --no-cpu-baseline --client-conv-x3 1 > $OUT/c2_x3.json 2>> $OUT/c2.err
for f in lib x3; do python3 -c "
import json
d=json.load(open('$OUT/c2_$f.json'))
print('$f', {k:((v.get('graph') or {}).get('ms_per_step'), (v.get('eager') or {}).get('ms_per_step'), (v.get('graph') or v.get('eager') or {}).get('loss')) for k,v in d['clients'].items()})"; done
tail -n 5 $OUT/c2.err
