#!/bin/bash
# round 6, call 19: SQ counters + HBM traffic (PMC) of the con_w kernels at D = 256 / 512 / 768 and of the A1 kernels at N = 4096
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/call19; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
bash tools/pmc_sq.sh conw python3 $ROOT/tools/kernel_bench.py --cases a5,a5wide > $OUT/r6_sq_a5.json 2> $OUT/sq_a5.err
python3 - <<P
import json
d=json.load(open('$OUT/r6_sq_a5.json'))
for k,v in d.items():
    if 'wide' in k or 'bank_fwd' in k:
        print(k, {c:v.get(c) for c in ('GRBM_GUI_ACTIVE','SQ_BUSY_CYCLES','SQ_VALU_MFMA_BUSY_CYCLES','SQ_INSTS_MFMA','SQ_INSTS_VALU','SQ_INSTS_LDS','SQ_WAVE_CYCLES','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_WAIT_INST_LDS','SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE')})
P
bash tools/pmc_run.sh conw python3 $ROOT/tools/kernel_bench.py --cases a5,a5wide > $OUT/pmc_a5.log 2>&1; tail -n 25 $OUT/pmc_a5.log
bash tools/pmc_run.sh a1 python3 $ROOT/tools/kernel_bench.py --cases a1 > $OUT/pmc_a1.log 2>&1; tail -n 25 $OUT/pmc_a1.log
cp $ROOT/gpurun_out/pmc_conw*/*.json $OUT/ 2>/dev/null; cp $ROOT/gpurun_out/pmc_a1*/*.json $OUT/ 2>/dev/null; ls $OUT
