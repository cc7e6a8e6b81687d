#!/bin/bash
# clients: fp32 fused BatchNorm on channels_last encoders -- parity, A/B against the library's BatchNorm, config 2 lines
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_clients
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MIOPEN_LOG_LEVEL=1
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_bnorm.py tests/test_gpu_framework.py::test_image_client_layouts_train_the_same tests/test_gpu_framework.py::test_client_contrast_step_in_a_hip_graph_equals_eager tests/test_gpu_framework.py::test_one_communication_round tests/test_gpu_configs.py::test_config0_round_two_image_two_text_clients_batch32 > $OUT/tests.log 2>&1; tail -8 $OUT/tests.log | cut -c1-400
echo '--- fused fp32 BatchNorm on'; timeout 600 python tools/client_layout_probe.py 2>/dev/null | grep mode
echo '--- library BatchNorm (CFL_NO_BN_FP32=1)'; CFL_NO_BN_FP32=1 timeout 600 python tools/client_layout_probe.py 2>/dev/null | grep "fp32 channels_last"
timeout 1200 python bench.py --config 2 --steps 30 --warmup 5 > $OUT/r5_config2_line.json 2> $OUT/c2.err
timeout 600 python bench.py --config 2 --steps 30 --warmup 5 --round none --client-layout nchw --no-cpu-baseline > $OUT/r5_config2_nchw_line.json 2>> $OUT/c2.err
timeout 600 python bench.py --config 2 --steps 30 --warmup 5 --round none --client-bf16 1 --no-cpu-baseline > $OUT/r5_config2_bf16_optin_line.json 2>> $OUT/c2.err
python3 - <<'PY'
import json
for f in ('r5_config2_line.json','r5_config2_nchw_line.json','r5_config2_bf16_optin_line.json'):
    try:
        d=json.loads(open('gpurun_out/r5_clients/'+f).read().strip().splitlines()[-1])
        print(f, d['value'], d['dtype'], {k:(v['eager']['ms_per_step'], v.get('graph') and v['graph']['ms_per_step'], v['a3a4_share_of_step']['product_path']) for k,v in d['clients'].items()}, d['round'] and (d['round']['miniature_warm_up_round_s'], d['round']['phases_s_rank0']))
    except Exception as e: print(f, 'ERR', e)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace_client -o cl --output-format csv -- python $ROOT/bench.py --config 2 --round none --steps 20 --warmup 3 --no-cpu-baseline > $OUT/trace_client.log 2>&1
python3 - $OUT <<'PY'
import csv, sys, glob
f = (glob.glob(sys.argv[1]+'/trace_client/*kernel_stats.csv')+glob.glob(sys.argv[1]+'/trace_client/*/*kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(f)))
w = csv.writer(open(sys.argv[1]+'/r5_client_step_kernel_stats.csv','w'))
w.writerow(['Name','Calls','TotalDurationNs','AverageNs','Percentage','MinNs','MaxNs'])
for r in rows: w.writerow([r['Name'][:140], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
for r in rows[:16]: print(r['Name'][:80], r['Calls'], r['Percentage'])
PY
rm -rf $OUT/trace_client
