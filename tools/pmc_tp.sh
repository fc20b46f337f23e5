#!/bin/bash
# PMC passes over the boundary-A kernel (tp_col_kernel): where its waves' cycles go.   usage (GPU box): tools/pmc_tp.sh [layer]
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
L=${1:-3}
OUT=$ROOT/gpurun_out/pmc_tp_L$L
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -o tp -- python $ROOT/tools/bench_tp.py --layer $L --edges 800000 --iters 4 > "$OUT/pmc$i.log" 2>&1
done
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
res = {}
for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'tp_col_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        res[k] = sum(v) / len(v)
for k in sorted(res):
    print('%-28s %16.0f' % (k, res[k]))
E = 800000
if 'SQ_WAVE_CYCLES' in res:
    wc = res['SQ_WAVE_CYCLES']
    for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA'):
        if k in res:
            print('%-28s / WAVE_CYCLES = %.3f' % (k, res[k] / wc))
for k in ('SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_SMEM', 'SQ_INSTS'):
    if k in res:
        print('%-28s per edge = %.1f' % (k, res[k] / E))
if 'SQ_LDS_BANK_CONFLICT' in res and 'SQ_LDS_IDX_ACTIVE' in res:
    print('LDS bank conflict cycles / LDS active cycles = %.3f' % (res['SQ_LDS_BANK_CONFLICT'] / res['SQ_LDS_IDX_ACTIVE']))
if 'SQ_LEVEL_WAVES' in res and 'SQ_BUSY_CYCLES' in res:
    print('mean waves in flight per SQ-busy cycle (chip) = %.1f' % (res['SQ_LEVEL_WAVES'] / res['SQ_BUSY_CYCLES']))
PY
