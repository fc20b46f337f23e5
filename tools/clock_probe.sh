#!/bin/bash
# Effective shader clock of the conv kernels (GRBM_GUI_ACTIVE / kernel duration) and their matrix-pipe busy share, for conv_kernel 0 and 2 (DDK_CONV_KERNEL) on fixed inputs
# (y = 1 needs a VARIANT library with tools/variants/k_conv_y.hip: tools/build_variant_y.sh y, then DDK_LIB=$(pwd)/ab_libs/libddk_y.so tools/clock_probe.sh ...; the product library refuses conv_kernel = 2)
# (tools/conv_fixed.py).  Run on the GPU box:  tools/clock_probe.sh <out-name>
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-clock}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
for y in 0 1; do
  DDK_CONV_KERNEL=$((2 * y)) rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/y$y" -o p -- python $ROOT/tools/conv_fixed.py --reps 4 > "$OUT/y$y.log" 2>&1
done
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
res = {}
for y in (0, 1):
    d = os.path.join(out, f'y{y}')
    dur = {}
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'conv_' in r['Kernel_Name'] and 'det_fix' not in r['Kernel_Name']:
                dur[r['Dispatch_Id']] = (r['Kernel_Name'][:40], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    cnt = defaultdict(dict)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Dispatch_Id'] in dur:
                cnt[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
    big = [(k, v) for k, v in dur.items() if v[1] > 300000 and 'GRBM_GUI_ACTIVE' in cnt[k]]      # the long launches (layers 1-3 at t = 1.0 / 0.6)
    ns = sum(v[1] for _, v in big); gui = sum(cnt[k]['GRBM_GUI_ACTIVE'] for k, _ in big); mf = sum(cnt[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0) for k, _ in big)
    res[f'conv_kernel={2 * int(y)}'] = {'kernel': big[0][1][0] if big else None, 'launches': len(big), 'mean_us': ns / max(len(big), 1) / 1e3,
                              'effective_clock_GHz': gui / max(ns, 1), 'mfma_busy_share_of_cycles_x_simds': mf / max(gui, 1) / 1024 * 8 if False else mf / max(gui * 1024 / 8, 1)}
json.dump(res, open(os.path.join(out, 'summary.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
find "$OUT" -name "*.csv" -delete
