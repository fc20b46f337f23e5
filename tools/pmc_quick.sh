#!/bin/bash
# Quick PMC passes over the bench's conv kernel (run on the GPU box through gpurun): where the waves' cycles go.
#   usage: tools/pmc_quick.sh <out-name> ["bench args"]
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
NAME=${1:-pmc}
ARGS=${2:---steps 4 --warmup 1 --no-cpu-baseline --no-alt --no-device-loop --no-extras}
OUT=$ROOT/gpurun_out/$NAME
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "SQ_INSTS SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -o bench -- python $ROOT/bench.py $ARGS > "$OUT/pmc$i.log" 2>&1
done
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
res = defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, 'pmc*'))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'conv_' in k:
                acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, cs in acc.items():
            for c, v in cs.items():
                res[k][c] = sum(v) / len(v)
                res[k]['launches'] = len(v)
json.dump(res, open(os.path.join(out, 'summary.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
find "$OUT" -name "*.csv" -delete
