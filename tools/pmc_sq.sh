#!/bin/bash
# SQ / LDS / MFMA counters of the hand-written kernels of a command (one rocprofv3 --pmc pass per counter group).
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/g$i -o pmc --output-format csv -- "$@" > $OUT/g$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys, json
out = {}
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace('void ', '')
        if 'cfl_' not in name: continue
        a = out.setdefault(name, {}).setdefault(r['Counter_Name'], [0, 0.0])
        a[0] += 1; a[1] += float(r['Counter_Value'])
print(json.dumps({k: {c: round(v[1] / v[0], 1) for c, v in sorted(d.items())} for k, d in sorted(out.items())}, indent=1))
PY
