# round-end evidence beyond the driver's command: boundary-A kernel profile, configs 3 / 4 / 5, the 363-complex stream (config 4)
mkdir -p gpurun_out/r6f
bash tools/profile_tp.sh > gpurun_out/r6f/profile_tp.log 2>&1; tail -5 gpurun_out/tp_prof/summary.md
for c in 3 4 5; do
  python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-device-loop --no-timesplit > gpurun_out/r6f/bench_config${c}_n1.json 2> gpurun_out/r6f/bench_config${c}.err
  python tools/show_bench.py gpurun_out/r6f/bench_config${c}_n1.json | head -4
done
python bench.py --config 4 --complexes 363 --steps 363 --warmup 2 --no-cpu-baseline --no-alt --no-device-loop --no-timesplit > gpurun_out/r6f/bench_config4_363.json 2> gpurun_out/r6f/bench_config4_363.err
python tools/show_bench.py gpurun_out/r6f/bench_config4_363.json | head -4
