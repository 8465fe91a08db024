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


# ---- sample statistics over sharded flow samples (dist.sharded_flow_corrs / sharded_mean_motion_map) ----------------------
def _flow_cpu_hooks():
    from oracle import flowstats_oracle as FO

    def cov_rows(x, row0, nrows, use_cov):
        out = []
        for b in range(x.shape[0]):
            c = torch.cov(x[b]) if use_cov else torch.corrcoef(x[b])
            c[torch.isnan(c)] = 0
            out.append(c[row0:row0 + nrows])
        return torch.stack(out, 0)

    def sum_fn(fl, nps, eps):
        return FO.compute_flow_samples_magnitude(fl, normalize=nps, eps=eps).sum(-1)

    def finish_fn(total, n, normalize, eps):
        m = total / n
        return FO.compute_mean_motion_map(m, eps=eps) if normalize else m

    return FO.flow_features, cov_rows, sum_fn, finish_fn


def _flow_inputs(S):
    g = torch.Generator().manual_seed(5)
    return torch.randn(2, 2, 8, 8, S, generator=g) * 2


def _flow_worker(rank, world, port, S, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    feats, cov_rows, sum_fn, finish_fn = _flow_cpu_hooks()
    lo, hi = cdist.shard_range(S, rank, world)
    local = _flow_inputs(S)[..., lo:hi].contiguous()
    cov = cdist.sharded_flow_corrs(local, downsample=2, use_covariance=True, gather=True, features_fn=feats, cov_rows_fn=cov_rows)
    slab = cdist.sharded_flow_corrs(local, downsample=2, use_covariance=False, gather=False, features_fn=feats, cov_rows_fn=cov_rows)
    mm = cdist.sharded_mean_motion_map(local, normalize_per_sample=True, normalize=True, sum_fn=sum_fn, finish_fn=finish_fn)
    torch.save((cov, slab, mm), os.path.join(out_dir, "f%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("S", [12, 7, 1])
def test_sharded_flow_statistics_match_single_process(tmp_path, S):
    from oracle import flowstats_oracle as FO

    world = 2
    mp.spawn(_flow_worker, args=(world, _free_port(), S, str(tmp_path)), nprocs=world, join=True)
    fl = _flow_inputs(S)
    cov = FO.compute_flow_corrs(fl, 2, True).reshape(2, 16, 16)
    corr = FO.compute_flow_corrs(fl, 2, False).reshape(2, 16, 16)
    mm = FO.compute_mean_motion_map(fl, normalize_per_sample=True, normalize=True)
    for r in range(world):
        c, slab, m = torch.load(os.path.join(str(tmp_path), "f%d.pt" % r))
        lo, hi = cdist.shard_range(16, r, world)
        assert torch.allclose(c, cov, atol=1e-6) and torch.allclose(slab, corr[:, lo:hi], atol=1e-6)
        assert torch.allclose(m, mm, atol=1e-6)
