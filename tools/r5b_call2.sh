#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5b2
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
run() { name=$1; shift; ( timeout 300 "$@" ) > $OUT/$name.log 2>&1; echo "== $name rc=$?"; grep -E "passed|failed|error|Fatal|Error" $OUT/$name.log | head -5 | cut -c1-300; }
run adamp_toy python -m pytest tests/test_gpu_optimizer.py -m gpu -q -x -k "graph or stale"
run mm_threadlocal python -m pytest tests/test_gpu_framework.py -m gpu -q -x -k "mm_client_contrast_step_in_a_hip_graph"
CFL_GRAPH_CAPTURE_MODE=global run mm_global python -m pytest tests/test_gpu_framework.py -m gpu -q -x -k "mm_client_contrast_step_in_a_hip_graph"
CFL_GRAPH_CAPTURE_MODE=relaxed run mm_relaxed python -m pytest tests/test_gpu_framework.py -m gpu -q -x -k "mm_client_contrast_step_in_a_hip_graph"
CFL_NO_TWO_STREAM=1 CFL_GRAPH_CAPTURE_MODE=global run mm_global_onestream python -m pytest tests/test_gpu_framework.py -m gpu -q -x -k "mm_client_contrast_step_in_a_hip_graph"
CFL_GRAPH_CAPTURE_MODE=global run server_global python -m pytest tests/test_gpu_framework.py -m gpu -q -x -k "server_contrastive_step_in_a_hip_graph or dropout_masks"
