cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/call28
timeout 600 python -m pytest tests/test_gpu_x3conv.py -q 2>&1 | tail -2
timeout 900 python tools/kernel_bench.py --cases x3conv > gpurun_out/call28/r6_x3conv_late_dma.jsonl 2> gpurun_out/call28/kb.err
python3 - <<P
import json
for l in open('gpurun_out/call28/r6_x3conv_late_dma.jsonl'):
    d=json.loads(l); print(d['case'], {k:v for k,v in d.items() if k.endswith('fwd_us')})
P
