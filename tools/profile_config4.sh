#!/bin/bash
# Per-kernel time of the north-star stream (config 4: DisCo score model + AR passes + confidence model, timesplit-shaped receptors), per complex.
# Run on the GPU box through gpurun; writes gpurun_out/c4prof/summary.md.
#   usage: tools/profile_config4.sh [complexes]
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
N=${1:-48}
OUT=/tmp/c4prof
rm -rf "$OUT"; mkdir -p "$OUT" "$ROOT/gpurun_out/c4prof"
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o bench -- python $ROOT/bench.py --config 4 --complexes $N --steps $N --warmup 2 \
    --no-cpu-baseline --no-alt --no-device-loop --no-timesplit --no-tp-boundary > "$OUT/run.log" 2> "$OUT/run.err"
tail -3 "$OUT/run.err"
python - "$OUT" "$N" > "$ROOT/gpurun_out/c4prof/summary.md" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out, n = sys.argv[1], int(sys.argv[2])
line = [l for l in open(os.path.join(out, 'run.log')) if l.startswith('{')][-1]
d = json.loads(line)
f = glob.glob(os.path.join(out, '**', '*kernel_trace.csv'), recursive=True)[0]
agg = defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
calls = max(len(v) for k, v in agg.items() if 'conf_head_kernel' in k)        # sampling() calls of the whole run: the headline bracket and the extra brackets, warm-ups included (one confidence head launch each)
tot = sum(sum(v) for v in agg.values())
print('# rocprofv3 --kernel-trace -- python bench.py --config 4 --complexes %d --steps %d --warmup 2 (MI355X; per sampling() call = per complex, %d calls over all brackets of the run)\n' % (n, n, calls))
print('under the profiler: %.2f complexes/s, %.2f ms per complex; kernel time per complex %.2f ms\n' % (d['value'], d['ms_per_step'], tot / calls / 1e3))
print('| kernel | launches per complex | avg us | min us | max us | us per complex | % of kernel time |\n|---|---|---|---|---|---|---|')
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print('| %s | %.1f | %.1f | %.1f | %.1f | %.1f | %.2f |' % (k[:100], len(v) / calls, sum(v) / len(v), min(v), max(v), sum(v) / calls, 100 * sum(v) / tot))
grp = defaultdict(float)
for k, v in agg.items():
    g = 'f16-limb conv kernel (score + AR + confidence layers)' if ('conv_x3_kernel<true' in k or 'conv_x2_kernel<true' in k) else ('confidence-only kernels (conf_*)' if 'conf_' in k else
        ('AR logits / decode' if 'ar_' in k else ('runtime fills / copies' if '__amd_rocclr' in k else ('other ddk kernels' if 'ddk::' in k else 'torch / other'))))
    grp[g] += sum(v)
print('\n| group | us per complex | % |\n|---|---|---|')
for g, t in sorted(grp.items(), key=lambda kv: -kv[1]):
    print('| %s | %.1f | %.1f |' % (g, t / calls, 100 * t / tot))
PY
cat "$ROOT/gpurun_out/c4prof/summary.md" | head -60
