#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5b4
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
timeout 800 python tools/mm_graph_probe.py > $OUT/mm_graph_probe.jsonl 2> $OUT/probe.err
cat $OUT/mm_graph_probe.jsonl | cut -c1-300
CFL_NO_TWO_STREAM=1 timeout 300 python tools/mm_graph_probe.py --variants "imgonly,full" | cut -c1-300 | tee $OUT/onestream.jsonl
AMD_LOG_LEVEL=3 timeout 200 python tools/mm_graph_probe.py --child --off "imgonly" 2>&1 >/dev/null | grep -E "hipStreamWaitEvent|hipEventRecord|Capture|hipGraph\]|hipStreamIsCapturing|hipStreamCreate|hipEventCreate|hipMemcpy|hipMemset|hipMalloc|hipFree" | grep -v "KernelNode\|LaunchKernel" | tail -150 | cut -c1-260 > $OUT/imgonly_hip_tail.txt
tail -40 $OUT/imgonly_hip_tail.txt
