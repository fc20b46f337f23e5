#!/bin/bash
# rocprofv3 evidence for the HBM-bound boundary-A kernel (tp_col_kernel, k_tp.hip): kernel trace + FETCH_SIZE / WRITE_SIZE passes over tools/bench_tp.py
# (E = 800 000, layers 3, 1, 0) -> gpurun_out/tp_prof/summary.md   (run on the GPU box through gpurun; copy the summary to profiles/)
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/tp_prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
for L in 3 2 1 0; do
  CMD="python $ROOT/tools/bench_tp.py --layer $L --edges 800000 --iters 10"
  python $ROOT/tools/bench_tp.py --layer $L --edges 800000 --iters 10 --json $OUT/bench_L$L.json > "$OUT/plain_L$L.log" 2>&1      # (no profiler: the events-timed figure of the last column)
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_L$L" -o tp -- $CMD > "$OUT/trace_L$L.log" 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_L$L" -o tp -- $CMD > "$OUT/fetch_L$L.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write_L$L" -o tp -- $CMD > "$OUT/write_L$L.log" 2>&1
done
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
def find(d, suffix):
    r = glob.glob(os.path.join(d, '**', '*' + suffix), recursive=True)
    return r[0] if r else None
lines = ['# tp_col_kernel (k_tp.hip): FasterTensorProduct.forward at the reference op boundary, weights [E, W] in HBM (SURVEY 8(d) boundary A)', '',
         'rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python tools/bench_tp.py --layer L --edges 800000 --iters 10', '',
         '| layer | W | algorithmic B/edge | avg launch us (kernel trace) | GB/s algorithmic | / 8000 | / 6300 | FETCH_SIZE KB (x2: gfx950 correction) | WRITE_SIZE KB | counter B/edge | counter / algorithmic | bench_tp.py alone (events, no profiler) ms |',
         '|---|---|---|---|---|---|---|---|---|---|---|---|']
for L in (3, 2, 1, 0):
    b = json.load(open(os.path.join(out, f'bench_L{L}.json')))
    E, alg = b['edges'], b['algorithmic_bytes_per_edge']
    rows = [r for r in csv.DictReader(open(find(os.path.join(out, f'trace_L{L}'), 'kernel_trace.csv'))) if 'tp_col_kernel' in r['Kernel_Name']]
    us = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
    us = us[1:] if len(us) > 1 else us
    avg = sum(us) / len(us)
    def pmc(kind, name):
        f = find(os.path.join(out, f'{kind}_L{L}'), 'counter_collection.csv')
        v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'tp_col_kernel' in r['Kernel_Name'] and r['Counter_Name'] == name]
        return sum(v) / max(len(v), 1)
    fe, wr = pmc('fetch', 'FETCH_SIZE'), pmc('write', 'WRITE_SIZE')
    cb = (2 * fe + wr) * 1024 / E
    gbs = alg * E / (avg * 1e-6) / 1e9
    lines.append(f'| {L} | {b["W"]} | {alg} | {avg:.1f} ({len(us)} launches) | {gbs:.0f} | {gbs / 8000:.3f} | {gbs / 6300:.3f} | {fe:.0f} | {wr:.0f} | {cb:.0f} | {cb / alg:.3f} | {b["ms_per_call"]:.4f} |')
open(os.path.join(out, 'summary.md'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -delete
