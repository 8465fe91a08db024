mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2a/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench_base8.json 2> gpurun_out/r2a/bench_base8.err
python bench.py --workload prompts256 --steps 5 --warmup 2 > gpurun_out/r2a/bench_prompts.json 2>&1
python bench.py --workload large4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2a/bench_large4.json 2>&1
python bench.py --workload imu4 --steps 5 --warmup 2 > gpurun_out/r2a/bench_imu4.json 2>&1
python tools/latency_b1.py > gpurun_out/r2a/latency_b1.log 2>&1
cat gpurun_out/r2a/pytest.log
