import os, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world):
    os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]="29533"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.full((4,), float(rank), device="cuda:0")
    try:
        dist.broadcast(t, src=1); print(rank, "broadcast ok", t.tolist())
        out = torch.empty(8, device="cuda:0"); dist.all_gather_into_tensor(out, torch.full((4,), float(rank), device="cuda:0")); print(rank, "all_gather_into_tensor ok", out.tolist())
        dist.all_reduce(t); print(rank, "all_reduce ok", t.tolist())
    except Exception as e:
        print(rank, "FAILED", type(e).__name__, str(e)[:200])
    dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(w, args=(2,), nprocs=2, join=True)
