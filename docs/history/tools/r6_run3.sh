#!/bin/bash
# round 6, GPU call 3: wgrad3x3 v2 (all four map sizes, pipelined fragment reads): parity, stand-alone timing, SQ counters, in-step A/B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_run3
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
( timeout 900 python -m pytest tests/test_gpu_wgrad.py -x -q -m gpu ) > $OUT/test_wgrad.log 2>&1
tail -15 $OUT/test_wgrad.log
timeout 600 python tools/kernel_bench.py --cases wgrad3 > $OUT/r6_wgrad3_kernel_bench.jsonl 2> $OUT/kb.err
cat $OUT/r6_wgrad3_kernel_bench.jsonl; tail -n 3 $OUT/kb.err
timeout 900 python tools/ab_step.py --knob wgrad3 --rounds 6 > $OUT/r6_ab_wgrad3.json 2> $OUT/ab.err
cat $OUT/r6_ab_wgrad3.json; tail -n 3 $OUT/ab.err
( timeout 900 python -m pytest tests/test_gpu_gru.py tests/test_gpu_optimizer.py -x -q -m gpu ) > $OUT/test_opt_gru.log 2>&1
tail -n 5 $OUT/test_opt_gru.log
( timeout 1500 python -m pytest tests/test_gpu_framework.py -x -q -m gpu ) > $OUT/test_framework.log 2>&1
tail -n 8 $OUT/test_framework.log
bash tools/pmc_sq.sh w3 python $ROOT/tools/kernel_bench.py --cases wgrad3 > $OUT/r6_sq_wgrad3.json 2> /dev/null
python3 -c "
import json
d=json.load(open('$OUT/r6_sq_wgrad3.json'))
print(json.dumps({k:v for k,v in d.items() if 'conv3x3' in k}, indent=1))"
