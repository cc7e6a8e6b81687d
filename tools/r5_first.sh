#!/bin/bash
# round 5, first GPU call: the new tests + bench --config 2 at full size + the default bench line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_first
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
timeout 1500 python -m pytest -x -q -m gpu tests/test_bench_launch.py::test_bench_config2_two_ranks_gloo_runs_the_client_round \
   tests/test_gpu_parity.py::test_integration_md_binding_stubs_run_against_the_oracle \
   tests/test_gpu_framework.py::test_bench_forward_flops_counts_both_towers \
   tests/test_gpu_framework.py::test_client_contrast_step_in_a_hip_graph_equals_eager \
   tests/test_gpu_framework.py::test_one_communication_round \
   tests/test_gpu_multirank.py > $OUT/tests.log 2>&1
tail -30 $OUT/tests.log
timeout 900 python bench.py --config 2 --steps 30 --warmup 5 > $OUT/config2_line.json 2> $OUT/config2.err
tail -5 $OUT/config2.err
cat $OUT/config2_line.json | cut -c1-3000
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err
cat $OUT/bench_line.json | cut -c1-1500
