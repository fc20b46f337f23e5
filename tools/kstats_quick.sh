#!/bin/bash
# Per-kernel durations of a short bench run (run on the GPU box through gpurun): the non-conv kernels of a reverse step at a glance.
#   usage: tools/kstats_quick.sh ["bench args"]
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
ARGS=${1:---steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-device-loop --no-extras}
OUT=/tmp/kstats
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o bench -- python $ROOT/bench.py $ARGS > "$OUT/run.log" 2>&1
grep '^{' "$OUT/run.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('value', round(d['value'],2), 'ms_per_step', round(d['ms_per_step'],2), '(under the profiler)')"
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)[0]
agg = defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
fw = len(agg[[k for k in agg if 'graph_fill' in k][0]])
print('forwards', fw)
tot_nc = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if 'ddk::' not in k:
        continue
    per = sum(v) / fw
    if 'conv_x3_kernel<true' not in k and 'conv_x2_kernel<true' not in k:
        tot_nc += per
    print('%-70s calls/fw %5.2f  avg %7.1f us  min %6.1f  per forward %7.1f us' % (k[:70], len(v) / fw, sum(v) / len(v), min(v), per))
print('non-conv ddk kernels per forward: %.1f us' % tot_nc)
PY
