#!/bin/bash
# development aid: tools/bench_tp.py over the kernel variants of k_tp.hip (DDK_TP_VARIANT / DDK_TP_GRID) -> gpurun_out/tp_sweep.log
mkdir -p gpurun_out
: > gpurun_out/tp_sweep.log
for L in 3 1 0 2; do
  for V in ${VARIANTS:-0 1 2 3 4 5 6}; do
    for G in ${GRIDS:-0}; do
      echo "== layer $L variant $V grid $G" >> gpurun_out/tp_sweep.log
      DDK_TP_VARIANT=$V DDK_TP_GRID=$G timeout 300 python tools/bench_tp.py --layer $L --edges 800000 --iters 10 >> gpurun_out/tp_sweep.log 2>&1
    done
  done
done
grep -E "^==|ms_per_call|Error|error|assert" gpurun_out/tp_sweep.log | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('=='): print(l, end=' ')
    elif l.startswith('{'):
        d = json.loads(l); print('%.4f ms %.0f GB/s err %.1e' % (d['ms_per_call'], d['achieved_GBps_algorithmic'], d['max_rel_err_vs_fp64']))
    else: print(l)
"
