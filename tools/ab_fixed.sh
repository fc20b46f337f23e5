#!/bin/bash
# same-box A/B of library builds on FIXED inputs (tools/conv_fixed.py): tools/ab_fixed.sh <reps> <name1> <name2> ...   (ab_libs/libddk_<name>.so)
REPS=$1; shift
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    echo -n "$v  "
    DDK_LIB=$(pwd)/ab_libs/libddk_$v.so timeout 300 python tools/conv_fixed.py --reps 10 2>/dev/null | grep -E "^mean" 
  done
done
