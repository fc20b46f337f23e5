# usage: run_variants.sh name[:ENV=VAL] ...
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
  for l in 3 1; do
    echo -n "$spec: "; env $envs DDK_LIB=disco_diffdock_amd/variants/libddk_$v.so python tools/bench_conv.py --layer $l --edges 800000 --iters 20 2>&1 | tail -1
  done
done
