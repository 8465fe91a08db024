"""Probe (GPU box): can two processes that share ONE GPU form an RCCL communicator through the C ABI?  (NCCL refuses duplicate devices;
if RCCL does too, the multi-rank RCCL path can only be exercised on a multi-GPU node.)  Run under `timeout`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29544"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from counterfactualworldmodels_amd import dist as cdist
    try:
        comm = cdist.get_comm(torch.device("cuda:0"))
        print(rank, type(comm).__name__, "initialised", flush=True)
        buf = torch.full((1024,), rank + 1, dtype=torch.uint8, device="cuda:0")
        comm.broadcast_bytes(buf, 0); torch.cuda.synchronize()
        print(rank, "broadcast ->", int(buf[0]), flush=True)
        local = torch.full((3, 4), float(rank), device="cuda:0"); out = torch.empty(6, 4, device="cuda:0")
        comm.all_gather_blocks(local, out, [3, 3]); torch.cuda.synchronize()
        print(rank, "allgatherv ->", out[:, 0].tolist(), flush=True)
    except Exception as e:
        print(rank, "FAILED:", type(e).__name__, str(e)[:300], flush=True)
    dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(w, args=(2,), nprocs=2, join=True)
