#!/bin/bash
# round 6, GPU call 2: the new 3x3 weight-gradient kernel (parity, stand-alone timing, in-step A/B) + the ADVICE fixes' GPU tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_run2
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
( timeout 900 python -m pytest tests/test_gpu_wgrad.py -x -q -m gpu ) > $OUT/test_wgrad.log 2>&1
tail -15 $OUT/test_wgrad.log
( timeout 900 python -m pytest tests/test_gpu_optimizer.py tests/test_gpu_gru.py -x -q -m gpu ) > $OUT/test_opt_gru.log 2>&1
tail -5 $OUT/test_opt_gru.log
timeout 600 python tools/kernel_bench.py --cases wgrad3 > $OUT/r6_wgrad3_kernel_bench.jsonl 2> $OUT/kb.err
cat $OUT/r6_wgrad3_kernel_bench.jsonl; tail -3 $OUT/kb.err
timeout 900 python tools/ab_step.py --knob wgrad3 --rounds 6 > $OUT/r6_ab_wgrad3.json 2> $OUT/ab.err
cat $OUT/r6_ab_wgrad3.json
timeout 900 python tools/ab_step.py --knob w3split16 --rounds 4 > $OUT/r6_ab_w3split16.json 2>> $OUT/ab.err
cat $OUT/r6_ab_w3split16.json
tail -3 $OUT/ab.err
( timeout 1500 python -m pytest tests/test_gpu_framework.py -x -q -m gpu -k "graph or client or round" ) > $OUT/test_framework_graphs.log 2>&1
tail -5 $OUT/test_framework_graphs.log
