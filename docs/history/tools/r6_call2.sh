#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call2; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
export MIOPEN_LOG_LEVEL=1
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-client-steps > $OUT/bench_fresh_$i.json 2> $OUT/bench_fresh_$i.err
  python3 -c "import json;d=json.load(open('$OUT/bench_fresh_$i.json'));print('fresh',$i,d['ms_per_step'],d['value'])"
done
( time timeout 900 python -m pytest tests/test_gpu_x3conv.py -q -x ) > $OUT/x3_tests.log 2>&1; tail -n 15 $OUT/x3_tests.log
timeout 900 python tools/kernel_bench.py --cases x3conv > $OUT/x3conv_probe.jsonl 2> $OUT/x3conv_probe.err; tail -n 3 $OUT/x3conv_probe.err
python3 - <<'PY'
import json,os
for ln in open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/call2/x3conv_probe.jsonl'):
    if ln.startswith('{'):
        d=json.loads(ln); print(d['case'], {k:v for k,v in d.items() if 'wgrad' in k})
PY
( time timeout 1200 python -m pytest tests/test_gpu_framework.py -q -x ) > $OUT/fw_tests.log 2>&1; tail -n 5 $OUT/fw_tests.log
python bench.py --no-cpu-baseline --no-client-steps > $OUT/bench_after_tests.json 2> $OUT/bench_after_tests.err
python3 -c "import json;d=json.load(open('$OUT/bench_after_tests.json'));print('after tests',d['ms_per_step'],d['value'])"
