# same-box A/B of the boundary-A kernel: ab_libs/libddk_tphead.so (the committed kernel) against the working tree's library
for rep in 1 2 3; do
for L in 3 2 1 0; do
  for V in HEAD new; do
    if [ $V = HEAD ]; then export DDK_LIB=$(pwd)/ab_libs/libddk_tphead.so; else unset DDK_LIB; fi
    echo "== $V layer $L"; python tools/bench_tp.py --layer $L --edges 800000 --iters 10 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['achieved_GBps_algorithmic']), d['max_rel_err_vs_fp64'])"
  done
done; done
