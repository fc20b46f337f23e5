mkdir -p gpurun_out/r6f
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r6f/bench_n1.json 2> gpurun_out/r6f/bench_n1.err ) 2> gpurun_out/r6f/bench_n1.time
tail -c 600 gpurun_out/r6f/bench_n1.err; cat gpurun_out/r6f/bench_n1.time
bash tools/profile_round.sh > gpurun_out/r6f/profile_round.log 2>&1
ls gpurun_out/prof_summary
