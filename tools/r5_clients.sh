#!/bin/bash
# channels_last client encoders: parity test, new find-db records, config 2 lines (fp32 channels_last = default; bf16 opt-in companion)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_clients
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_framework.py::test_image_client_layouts_train_the_same tests/test_gpu_framework.py::test_client_contrast_step_in_a_hip_graph_equals_eager tests/test_gpu_framework.py::test_one_communication_round > $OUT/tests.log 2>&1; tail -15 $OUT/tests.log | cut -c1-300
bash tools/record_find_db.sh > $OUT/record.log 2>&1; tail -6 $OUT/record.log
