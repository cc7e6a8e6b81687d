#!/bin/bash
# round 6, call 13: the ViT trunk on the fused pre-LN glue: op parity, chain parity, configs[4] line fused vs aten (batch 256 and 64) + kernel stats
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call13; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
export MIOPEN_LOG_LEVEL=1
( time timeout 900 python -m pytest tests/test_gpu_bert.py -q ) > $OUT/bert_tests.log 2>&1; tail -n 15 $OUT/bert_tests.log
( time timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_framework.py -q -k "config4" ) > $OUT/c4_tests.log 2>&1; tail -n 6 $OUT/c4_tests.log
timeout 900 python tools/config4_bench.py --batch 256 > $OUT/r6_config4_b256_line.json 2> $OUT/c4.err; cat $OUT/r6_config4_b256_line.json; tail -n 2 $OUT/c4.err
CFL_NO_VIT_FUSE=1 timeout 900 python tools/config4_bench.py --batch 256 > $OUT/r6_config4_b256_line_aten.json 2> $OUT/c4a.err; cat $OUT/r6_config4_b256_line_aten.json; tail -n 2 $OUT/c4a.err
timeout 900 python tools/config4_bench.py --batch 64 > $OUT/r6_config4_line.json 2> $OUT/c4s.err; cat $OUT/r6_config4_line.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace_c4 -o c4 --output-format csv -- python3 $ROOT/tools/config4_bench.py --batch 256 --steps 5 --warmup 2 > $OUT/trace_c4.log 2>&1
cp $OUT/trace_c4/*kernel_stats.csv $OUT/r6_config4_kernel_stats.csv 2>/dev/null; rm -rf $OUT/trace_c4
head -n 30 $OUT/r6_config4_kernel_stats.csv | cut -c1-160
