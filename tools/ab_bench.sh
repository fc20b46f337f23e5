#!/bin/bash
# A/B comparison of library builds on the GPU box: tools/ab_bench.sh <reps> <name1> <name2> ...   (libraries ab_libs/libddk_<name>.so, git-ignored)
REPS=$1; shift
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    DDK_LIB=$(pwd)/ab_libs/libddk_$v.so timeout 300 python bench.py --steps 8 --warmup 2 --no-alt --no-cpu-baseline --no-extras --no-device-loop 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$v', round(d['value'],2), round(d['roofline']['avg_launch_ms'],4))"
  done
done
