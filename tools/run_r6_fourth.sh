OUT=gpurun_out/r6d
mkdir -p $OUT
timeout 900 python -m pytest tests/test_flowstats_gpu.py tests/test_prompts_gpu.py tests/test_kernels_gpu.py -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 300 python bench.py --workload flowstats --steps 10 > $OUT/bench_flowstats.json 2> $OUT/bench_flowstats.err; tail -3 $OUT/bench_flowstats.err
timeout 300 python tools/dist_costs.py > $OUT/dist_costs.log 2>&1
MODE=fast timeout 300 python tools/ab_step.py attn_kernel 1 0 > $OUT/ab_step_attn_kernel_fast.log 2>&1
MODE=fast timeout 300 python tools/ab_step.py attn_kernel 1 0 1 > $OUT/ab_step_attn_kernel_fast_lanes1.log 2>&1
