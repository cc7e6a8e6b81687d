#!/bin/bash
# round 6, GPU call 9: the packed text tower at the server's public batch (128, host-bound), wgrad tests after the gate
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_run9
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
( timeout 900 python -m pytest tests/test_gpu_wgrad.py -q -m gpu ) > $OUT/test_wgrad.log 2>&1
tail -n 3 $OUT/test_wgrad.log
timeout 600 python tools/ab_step.py --knob bertpack --rounds 4 --batch 128 > $OUT/r6_ab_bertpack_b128.json 2>> $OUT/ab.err
cat $OUT/r6_ab_bertpack_b128.json
timeout 600 python tools/ab_step.py --knob wgrad1 --rounds 4 --batch 128 > $OUT/r6_ab_wgrad1_b128.json 2>> $OUT/ab.err
cat $OUT/r6_ab_wgrad1_b128.json
timeout 600 python tools/ab_step.py --knob wgrad3 --rounds 4 --batch 128 > $OUT/r6_ab_wgrad3_b128.json 2>> $OUT/ab.err
cat $OUT/r6_ab_wgrad3_b128.json
tail -n 3 $OUT/ab.err
T="timeout 600 python tools/federation_step_trace.py --batches 50 --rounds 2"
$T --tag packed > $OUT/fed_packed.json 2>> $OUT/fed.err
CFL_NO_BERT_PACK=1 $T --tag padded > $OUT/fed_padded.json 2>> $OUT/fed.err
cat $OUT/fed_packed.json $OUT/fed_padded.json > $OUT/r6_federation_step_trace_packing.jsonl
python3 -c "
import json
for l in open('$OUT/r6_federation_step_trace_packing.jsonl'):
    d=json.loads(l)
    print(d['tag'], [(p['wall_ms_per_batch'], p['issue_ms_per_batch'], p['median_rest_ms'], p['p90_rest_ms'], p['mem_gb']['reserved']) for p in d['phases']])"
tail -n 3 $OUT/fed.err
