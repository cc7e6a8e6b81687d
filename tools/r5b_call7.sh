#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5b8
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
go() { tag=$1; shift; env "$@" timeout 300 python tools/server_graph_ab.py --batch 128 --passes 1 --tag "$tag" 2>> $OUT/err.txt | tee -a $OUT/r5_server_graph_env_ab.jsonl | cut -c1-500; }
go default CFL_X=0
go queues1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
go queues2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
go queues8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
go nopacket DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
go packet1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
go onestream CFL_NO_TWO_STREAM=1
tail -3 $OUT/err.txt | cut -c1-200
