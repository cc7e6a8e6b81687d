#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call3; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
export MIOPEN_LOG_LEVEL=1
( time timeout 900 python -m pytest tests/test_gpu_x3conv.py -q ) > $OUT/x3_tests.log 2>&1; tail -n 15 $OUT/x3_tests.log
timeout 900 python tools/kernel_bench.py --cases x3conv > $OUT/x3conv_probe.jsonl 2> $OUT/x3conv_probe.err; tail -n 3 $OUT/x3conv_probe.err
python3 - <<'PY'
import json,os
for ln in open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/call3/x3conv_probe.jsonl'):
    if ln.startswith('{'):
        d=json.loads(ln); print(d['case'], {k:v for k,v in d.items() if k.endswith('fwd_us') or k in ('x3_wgrad_us','library_wgrad_us','x3_dgrad_us_incl_weight_rotation','best_variant')})
PY
( time timeout 1200 python -m pytest tests/test_gpu_framework.py -q -k "layout or client" ) > $OUT/fw_tests.log 2>&1; tail -n 5 $OUT/fw_tests.log
timeout 1500 python bench.py --config 2 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/config2_line.json 2> $OUT/config2.err
python3 -c "
import json
d=json.load(open('$OUT/config2_line.json'))
print({k:(v.get('graph') or v.get('eager') or {}).get('ms_per_step') for k,v in d['clients'].items()})
print(d['round']['ms_per_public_batch'], d['round']['phases_s_rank0'])
print({k:v for k,v in d['clients']['img']['hip_kernels'].items() if 'conv' in k})"
tail -n 3 $OUT/config2.err
