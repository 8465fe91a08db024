# Multi-rank control flow of bench.py on a ONE-GPU box: N ranks share cuda:0, gloo carries the launcher group and (CWM_COMM=torch) the
# data-path collectives.  Throughput numbers from this are meaningless (the ranks time-share the GPU); what it checks is that the
# sharded 256-prompt loop, the packed broadcast, the gather and the max-over-ranks clock run to a valid JSON line.
N=${1:-2}
export CWM_BENCH_ONE_DEVICE=1 CWM_BENCH_BACKEND=gloo CWM_COMM=torch
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus $N --steps 3 --warmup 1 --no-secondary 2>/dev/null | tail -1
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus $N --steps 2 --warmup 1 --workload prompts256 2>/dev/null | tail -1
