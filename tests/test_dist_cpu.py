"""world_size-2 gloo test of the prompt-sharding driver (no GPU): the sharded result must equal the
single-process result for even and ragged prompt counts."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from counterfactualworldmodels_amd import dist as cdist  # noqa: E402


def _build(x, prompts):
    # toy prompt construction: per-prompt input = frame pair + prompt-dependent offset, mask from the prompt
    b = prompts.shape[0]
    xs = x.expand(b, -1, -1, -1, -1) + prompts[:, 0].float().view(b, 1, 1, 1, 1)
    ms = (torch.arange(8).view(1, 8) == prompts[:, 1].view(b, 1))
    return xs, ms


def _predict(xs, ms):
    return xs.mean(dim=(1, 2, 3, 4), keepdim=False).view(-1, 1) * 2.0 + ms.float() @ torch.arange(8.0).view(8, 1)


def _single(x, prompts, chunk):
    return cdist.sharded_counterfactual_predictions(x, prompts, _build, _predict, "cpu", chunk=chunk)


def _worker(rank, world, port, S, chunk, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(1, 2, 3, 8, 8, generator=g) if rank == 0 else None
    prompts = torch.stack([torch.arange(S, dtype=torch.int32), torch.arange(S, dtype=torch.int32) % 8], 1) if rank == 0 else None
    y = cdist.sharded_counterfactual_predictions(x, prompts, _build, _predict, "cpu", chunk=chunk)
    torch.save(y, os.path.join(out_dir, "y%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("S,chunk", [(16, 4), (13, 4), (1, 32), (3, 1)])
def test_sharded_prompts_match_single_process(tmp_path, S, chunk):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), S, chunk, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(1, 2, 3, 8, 8, generator=g)
    prompts = torch.stack([torch.arange(S, dtype=torch.int32), torch.arange(S, dtype=torch.int32) % 8], 1)
    ref = _single(x, prompts, chunk)
    assert ref.shape == (S, 1)
    for r in range(world):
        y = torch.load(os.path.join(str(tmp_path), "y%d.pt" % r))
        assert torch.equal(y, ref), (r, S, chunk)


def test_shard_range_partitions():
    for total in (0, 1, 7, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [cdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
