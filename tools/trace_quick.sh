#!/bin/bash
# rocprofv3 --kernel-trace --stats over the bench's timed region (run on the GPU box through gpurun): per-kernel table into gpurun_out/<name>.md
#   usage: tools/trace_quick.sh <name> ["bench args"]
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
NAME=${1:-trace}
ARGS=${2:---steps 8 --warmup 2 --no-cpu-baseline --no-alt --no-device-loop --no-extras}
OUT=/tmp/prof_$NAME
rm -rf "$OUT"; mkdir -p "$OUT" "$ROOT/gpurun_out"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python $ROOT/bench.py $ARGS > "$OUT/trace.log" 2>&1
cd "$ROOT"
python - "$OUT" "$ROOT/gpurun_out/$NAME.md" "$ARGS" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
src, out, args = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(os.path.join(src, 'trace', '**', '*kernel_trace.csv'), recursive=True)[0]
agg = defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in agg.values())
lines = [f'# rocprofv3 --kernel-trace --stats -- python bench.py {args} (MI355X)', '', f'total kernel time {tot / 1e3:.2f} ms', '',
         '| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---|---|---|---|---|---|']
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    lines.append(f'| {k[:100]} | {len(v)} | {sum(v) / 1e3:.2f} | {sum(v) / len(v):.1f} | {min(v):.1f} | {max(v):.1f} | {100 * sum(v) / tot:.1f} |')
open(out, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[:40]))
PY
