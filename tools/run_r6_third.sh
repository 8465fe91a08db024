OUT=gpurun_out/r6c
mkdir -p $OUT
timeout 1700 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 300 python bench.py --workload flowstats --steps 10 > $OUT/bench_flowstats.json 2> $OUT/bench_flowstats.err; tail -3 $OUT/bench_flowstats.err
timeout 300 python bench.py --workload prompt_build --steps 20 > $OUT/bench_prompt_build.json 2> $OUT/bench_prompt_build.err; tail -3 $OUT/bench_prompt_build.err
timeout 300 python tools/dist_costs.py > $OUT/dist_costs.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_base8.json 2> $OUT/bench_base8.err
timeout 400 bash tools/run_bench_2ranks_1gpu.sh > $OUT/bench_2ranks_1gpu.log 2>&1
