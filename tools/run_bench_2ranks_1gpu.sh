# Multi-rank control flow of bench.py on a ONE-GPU box: N ranks share cuda:0, gloo carries the launcher group and (CWM_COMM=torch) the
# data-path collectives.  Throughput numbers from this are meaningless (the ranks time-share the GPU); what it checks is that
# `python bench.py --gpus N` launches its own ranks (bench.py self_launch: a child torch.distributed.run, started before anything
# touches the GPU), that the sharded 256-prompt loop, the packed broadcast, the per-chunk gather and the max-over-ranks clock run to
# a valid JSON line with n_gpus == N, and that without the test hooks the same command refuses to run on a box with fewer GPUs.
N=${1:-2}
echo "# without the hooks: must fail loudly (exit 2) on a 1-GPU box"
timeout 120 python bench.py --gpus $N --steps 2 --warmup 1; echo "exit code $?"
export CWM_BENCH_ONE_DEVICE=1 CWM_BENCH_BACKEND=gloo CWM_COMM=torch
timeout 150 python bench.py --gpus $N --steps 3 --warmup 1 --no-secondary 2>/dev/null | tail -1
timeout 150 python bench.py --gpus $N --steps 2 --warmup 1 --workload prompts256 2>/dev/null | tail -1
