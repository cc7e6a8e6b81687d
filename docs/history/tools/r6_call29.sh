cd $GRAFT_REPO_ROOT; OUT=gpurun_out/call29; mkdir -p $OUT; export MIOPEN_LOG_LEVEL=1
timeout 600 python -m pytest tests/test_gpu_x3conv.py -q 2>&1 | tail -2
for v in late early late early; do
  if [ $v = early ]; then export CFL_X3_DMA_EARLY=1; else unset CFL_X3_DMA_EARLY; fi
  timeout 900 python bench.py --config 2 --round none --steps 30 --warmup 5 --no-cpu-baseline --only-kinds img,mm > $OUT/c2_$v.json 2> $OUT/c2_$v.err
  python3 -c "
import json
d=json.load(open('$OUT/c2_$v.json'))
print('$v', {k:((v.get('graph') or {}).get('ms_per_step'), v['eager']['ms_per_step']) for k,v in d['clients'].items()}, {k:(x['launches_per_step'], x['us_per_launch']) for k,x in d['clients']['img']['hip_kernels'].items() if 'conv3x3_x3_kernel' in k})"
done
