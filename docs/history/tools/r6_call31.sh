cd $GRAFT_REPO_ROOT; OUT=gpurun_out/call31; mkdir -p $OUT
for v in d a d a; do
  if [ $v = a ]; then export CFL_CONW_WIDE_RB=a; else unset CFL_CONW_WIDE_RB; fi
  timeout 600 python tools/kernel_bench.py --cases a5,a5wide 2> $OUT/kb_$v.err | grep -v "tile GEMM" | grep -E "D=256|D=512" | sed "s/^{/{\"dma\": \"$v\", /" >> $OUT/r6_a5_asm_dma_ab.jsonl
done
unset CFL_CONW_WIDE_RB
python3 - <<P
import json
for l in open('$OUT/r6_a5_asm_dma_ab.jsonl'):
    d=json.loads(l); print(d['dma'], d['case'], d['kernels_us'].get('cfl_bank_stream_kernel'))
P
CFL_CONW_WIDE_RB=a timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "a5 or conw" 2>&1 | tail -2
