mkdir -p gpurun_out/r6h
for c in 3 4 5; do
  python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-device-loop --no-timesplit > gpurun_out/r6h/bench_config${c}_n1.json 2> gpurun_out/r6h/bench_config${c}.err
  python tools/show_bench.py gpurun_out/r6h/bench_config${c}_n1.json | head -6
done
python bench.py --config 4 --complexes 363 --steps 363 --warmup 2 --no-cpu-baseline --no-device-loop --no-timesplit > gpurun_out/r6h/bench_config4_363.json 2> gpurun_out/r6h/bench_config4_363.err
python tools/show_bench.py gpurun_out/r6h/bench_config4_363.json | head -6
bash tools/profile_config4.sh 48 > gpurun_out/r6h/profile_config4.log 2>&1; tail -8 gpurun_out/c4prof/summary.md
bash tools/ranks8_check.sh gpurun_out/r6h/ranks8.json > gpurun_out/r6h/ranks8.log 2>&1; tail -5 gpurun_out/r6h/ranks8.log
