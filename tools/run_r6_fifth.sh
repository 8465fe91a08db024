OUT=gpurun_out/r6i
mkdir -p $OUT
timeout 900 python -m pytest tests/test_flowstats_gpu.py tests/test_prompts_gpu.py tests/test_kernels_gpu.py -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 300 python bench.py --workload flowstats --steps 10 --no-cpu-baseline > $OUT/bench_flowstats.json 2> $OUT/bench_flowstats.err; tail -3 $OUT/bench_flowstats.err
timeout 300 python tools/dist_costs.py > $OUT/dist_costs.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_flowstats -- python3 bench.py --workload flowstats --steps 5 --no-cpu-baseline > $OUT/stats_flowstats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_flowstats -- python3 bench.py --workload flowstats --steps 5 --no-cpu-baseline > $OUT/pmc_mfma_flowstats.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -3
