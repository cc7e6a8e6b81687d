#!/bin/bash
# round 6, GPU call 12: x3 convolutions in the clients: training parity (layout test), graph tests, config-2 round with / without
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_run12
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
( timeout 1500 python -m pytest tests/test_gpu_framework.py -q -m gpu -k "layouts or hip_graph or communication_round" ) > $OUT/test_fw.log 2>&1
tail -n 8 $OUT/test_fw.log
( CFL_X3CONV=1 timeout 1500 python -m pytest tests/test_gpu_framework.py tests/test_gpu_configs.py -q -m gpu -k "hip_graph or communication_round or client or config2" ) > $OUT/test_fw_x3.log 2>&1
tail -n 8 $OUT/test_fw_x3.log
timeout 1500 python bench.py --config 2 --no-cpu-baseline --client-conv-x3 1 > $OUT/r6_config2_line_x3.json 2> $OUT/config2.err
python3 -c "
import json
d=json.load(open('$OUT/r6_config2_line_x3.json'))
print({k:(v.get('graph') or v.get('eager') or {}).get('ms_per_step') for k,v in d['clients'].items()})
print(d['round']['ms_per_public_batch'], d['round']['phases_s_rank0'], d['round']['recall_1_after_round'])"
tail -n 3 $OUT/config2.err
