# round 6, first GPU pass: the GPU test suite, the fast-mode attention kernel sweep, the per-rank fixed costs of the sharded loop, GEMM micro-benchmarks
OUT=gpurun_out/r6a
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
ATTN_KERNELS=1,3 timeout 300 python tools/microbench.py attn_sweep > $OUT/microbench_attn_sweep.log 2>&1
timeout 300 python tools/microbench.py attn > $OUT/microbench_attn.log 2>&1
timeout 300 python tools/dist_costs.py > $OUT/dist_costs.log 2>&1
timeout 600 python tools/microbench.py gemm > $OUT/microbench_gemm_b8.log 2>&1
timeout 400 bash tools/run_bench_2ranks_1gpu.sh > $OUT/bench_2ranks_1gpu.log 2>&1
timeout 300 python bench.py --workload prompts256 --steps 8 --warmup 2 > $OUT/bench_prompts256.json 2>$OUT/bench_prompts256.err
tail -20 $OUT/microbench_attn_sweep.log $OUT/dist_costs.log
