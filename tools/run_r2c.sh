mkdir -p gpurun_out/r2c
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2c/pytest.log
python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "fold" 2>&1 | grep "ln fold" > gpurun_out/r2c/fold_err.log
python tools/ablate_ln.py > gpurun_out/r2c/ablate.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r2c/bench_base8.json 2> gpurun_out/r2c/bench_base8.err
python bench.py --workload large4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2c/bench_large4.json 2>&1
cat gpurun_out/r2c/pytest.log gpurun_out/r2c/fold_err.log gpurun_out/r2c/ablate.log
