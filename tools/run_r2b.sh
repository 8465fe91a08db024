mkdir -p gpurun_out/r2b
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2b/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2b/bench_base8.json 2> gpurun_out/r2b/bench_base8.err
python bench.py --workload prompts256 --steps 5 --warmup 2 > gpurun_out/r2b/bench_prompts.json 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2b/smoke.log 2>&1
cat gpurun_out/r2b/pytest.log; tail -3 gpurun_out/r2b/smoke.log
