"""world_size 2 / 4 gloo tests of the prompt-sharding driver (no GPU): packed broadcast, rectangularisation on rank 0 only,
row-range prediction and the all-gather into the pre-sized result must reproduce the single-process result for even, ragged,
single-prompt and empty-shard cases."""
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
from counterfactualworldmodels_amd.masking import RectangularizeMasks  # noqa: E402

NT = 8


FRAME_ROWS = []  # rows whose frames each build call produced (this process): rank 0 must not build the frames of rows it does not predict


def _build(x, prompts, frames=True):
    # toy prompt construction: per-prompt input = frame pair + prompt-dependent offset; masks with DIFFERENT masked counts per row
    b = prompts.shape[0]
    xs = None
    if frames:
        FRAME_ROWS.append(b)
        xs = x.expand(b, -1, -1, -1, -1) + prompts[:, 0].float().view(b, 1, 1, 1, 1)
    ms = torch.arange(NT).view(1, NT) <= (prompts[:, 1].view(b, 1) % 5 + 2)
    return xs, ms


def _rect(masks):
    # the product's rectangulariser: global torch RNG, in place, once for ALL rows (rank 0 only)
    r = RectangularizeMasks("min")
    masks = r(masks.clone())
    return masks, r.last_num_masked


CALLS = []


def _predict(xs, ms, n_masked, chunk):
    assert (ms.sum(-1) == n_masked).all()          # every rank sees rectangular rows and the right count
    outs = []
    for c0 in range(0, xs.shape[0], chunk):
        CALLS.append(min(chunk, xs.shape[0] - c0))
        xc, mc = xs[c0:c0 + chunk], ms[c0:c0 + chunk]
        outs.append(xc.mean(dim=(1, 2, 3, 4)).view(-1, 1) * 2.0 + mc.float() @ torch.arange(float(NT)).view(NT, 1))
    return torch.cat(outs, 0)


def _inputs(S):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(1, 2, 3, 8, 8, generator=g)
    prompts = torch.stack([torch.arange(S, dtype=torch.int32), (torch.arange(S, dtype=torch.int32) * 7) % 11], 1)
    return x, prompts


def _single(S, chunk):
    x, prompts = _inputs(S)
    torch.manual_seed(123)
    return cdist.sharded_counterfactual_predictions(x, prompts, _build, _rect, _predict, "cpu", chunk=chunk, comm=cdist.LocalComm())


def _worker(rank, world, port, S, chunk, hint, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, prompts = _inputs(S) if rank == 0 else (None, None)
    torch.manual_seed(123 if rank == 0 else 999 + rank)   # only rank 0's RNG may matter
    shapes = ((1, 2, 3, 8, 8), (S, 2), NT) if hint else None
    FRAME_ROWS.clear()
    y = cdist.sharded_counterfactual_predictions(x, prompts, _build, _rect, _predict, "cpu", chunk=chunk, shapes=shapes)
    assert isinstance(cdist.get_comm("cpu"), cdist.TorchComm)
    used = cdist.get_comm("cpu").last_collective
    lo, hi = cdist.shard_range(S, rank, world)
    # every rank -- rank 0 too -- builds frames for the rows it predicts and nothing else (an empty shard: one row for the trailing shape)
    assert FRAME_ROWS == ([hi - lo] if hi > lo else [1]), (rank, FRAME_ROWS)
    y_loc = cdist.sharded_counterfactual_predictions(x, prompts, _build, _rect, _predict, "cpu", chunk=chunk, gather=False, shapes=shapes)
    torch.save((y, y_loc, used), os.path.join(out_dir, "y%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,S,chunk,hint", [(2, 16, 4, True), (2, 13, 4, False), (2, 1, 32, True), (4, 16, 3, True), (4, 6, 32, False), (4, 3, 1, True),
                                                 (2, 16, 8, True), (4, 16, 32, False)])
def test_sharded_prompts_match_single_process(tmp_path, world, S, chunk, hint):
    mp.spawn(_worker, args=(world, _free_port(), S, chunk, hint, str(tmp_path)), nprocs=world, join=True)
    ref = _single(S, chunk)
    assert ref.shape == (S, 1)
    for r in range(world):
        y, y_loc, used = torch.load(os.path.join(str(tmp_path), "y%d.pt" % r))
        assert torch.equal(y, ref), (r, S, chunk)              # all S rows, prompt order, on every rank
        lo, hi = cdist.shard_range(S, r, world)
        assert y_loc.shape == (hi - lo, 1)                      # gather=False keeps the local block (possibly empty)
        # every layout -- one chunk or several, equal or ragged shards -- goes through the plain equal-block all-gather
        assert used == "all_gather_into_tensor", (used, world, S, chunk)


def _failing_worker(rank, world, port, hint, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, prompts = _inputs(6) if rank == 0 else (None, None)

    def bad_rect(masks):
        raise ValueError("rect failed on rank 0")

    shapes = ((1, 2, 3, 8, 8), (6, 2), NT) if hint else None
    try:
        cdist.sharded_counterfactual_predictions(x, prompts, _build, bad_rect, _predict, "cpu", chunk=4, shapes=shapes)
        res = "no error"
    except Exception as e:  # noqa: BLE001
        res = type(e).__name__
    # a wrong shape hint is caught on rank 0 before the collective, and reaches the peers the same way
    try:
        cdist.sharded_counterfactual_predictions(x, prompts, _build, _rect, _predict, "cpu", chunk=4, shapes=((1, 2, 3, 8, 8), (7, 2), NT))
        res2 = "no error"
    except Exception as e:  # noqa: BLE001
        res2 = type(e).__name__
    torch.save((res, res2), os.path.join(out_dir, "e%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("hint", [True, False])
def test_rank0_failure_reaches_every_rank(tmp_path, hint):
    """A failure on rank 0 before the packed broadcast must not leave the peers blocked in the collective."""
    mp.spawn(_failing_worker, args=(2, _free_port(), hint, str(tmp_path)), nprocs=2, join=True)
    assert torch.load(os.path.join(str(tmp_path), "e0.pt")) == ("ValueError", "ValueError")
    assert torch.load(os.path.join(str(tmp_path), "e1.pt")) == ("RemoteRankError", "RemoteRankError")


def test_packed_buffer_round_trip():
    x, prompts = _inputs(5)
    masks = torch.arange(NT).view(1, NT) < torch.tensor([[3]] * 5)
    buf = cdist.pack_inputs(x, prompts, masks, 3, "cpu")
    assert buf.dtype == torch.uint8 and buf.numel() % 16 == 0
    assert buf.numel() == cdist._layout(x.shape, prompts.shape, NT)[3]
    x2, p2, m2, n = cdist.unpack_inputs(buf)
    assert torch.equal(x2, x) and torch.equal(p2, prompts) and torch.equal(m2, masks) and n == 3 and m2.dtype == torch.bool
    bad = buf.clone()
    bad[0] ^= 0xFF
    with pytest.raises(AssertionError):
        cdist.unpack_inputs(bad)


def test_chunks_are_row_ranges_of_the_local_block():
    CALLS.clear()
    _single(10, 4)
    assert CALLS == [4, 4, 2]


def test_gather_pointer_arithmetic_for_every_rank_of_the_8_gpu_layout():
    """`RcclComm.gather_args` (the pointer arithmetic behind cwm_allgather / cwm_allgatherv) for every rank of the layouts the multi-GPU node will
    see first: 256 prompts in chunks of 32 on 8 ranks (one equal block per rank, back to back in the result -> ncclAllGather in place: send ==
    recv + rank * bytes, everything inside the result) and on 2 / 4 ranks (chunk c of every rank -> a [world, 32] staging block, the same in-place
    ncclAllGather, then `_place` moves block r to rows r * 256 / world + 32 c); ragged layouts keep the per-rank byte offsets of cwm_allgatherv
    (the flow-statistics gathers)."""
    row, base = 3 * 224 * 224 * 4, 0x7F0000000000
    for world in (2, 4, 8):
        bounds = [cdist.shard_range(256, r, world) for r in range(world)]
        n_chunks = -(-max(h - l for l, h in bounds) // 32)
        assert n_chunks == 8 // world
        for c in range(n_chunks):
            offs = [min(l + c * 32, h) for l, h in bounds]
            cnts = [min(l + (c + 1) * 32, h) - o for (l, h), o in zip(bounds, offs)]
            assert cnts == [32] * world
            for r in range(world):
                if world == 8:  # one chunk per rank: the blocks are back to back in the RESULT -> the ring collective, in place
                    kind, send, recv, nbytes = cdist.RcclComm.gather_args(base, row, offs, cnts, r)
                    assert recv + world * nbytes == base + 256 * row
                else:           # chunk c of every rank: the staging block [world, 32] (what `all_gather_stage` passes on)
                    kind, send, recv, nbytes = cdist.RcclComm.gather_args(base, row, [32 * q for q in range(world)], [32] * world, r)
                    assert recv + world * nbytes == base + world * 32 * row
                assert kind == "allgather" and nbytes == 32 * row
                assert send == recv + r * nbytes and recv == base
    for total, world in ((250, 8), (3, 8), (33, 4)):
        bounds = [cdist.shard_range(total, r, world) for r in range(world)]
        offs, cnts = [l for l, _ in bounds], [h - l for l, h in bounds]
        for r in range(world):
            kind, send, recv, o, n = cdist.RcclComm.gather_args(base, row, offs, cnts, r)
            assert kind == "allgatherv" and recv == base and n == [k * row for k in cnts] and o == [k * row for k in offs]
            assert (send is None) == (cnts[r] == 0) and (send is None or send == base + offs[r] * row)
            assert sum(n) == total * row and max(a + b for a, b in zip(o, n)) == total * row


@pytest.mark.parametrize("total,world,chunk", [(256, 2, 32), (256, 4, 32), (256, 8, 32), (250, 8, 32), (13, 2, 4), (3, 4, 1), (33, 4, 32), (16, 4, 3)])
def test_staged_gather_places_every_row(total, world, chunk):
    """The staging layout of the per-chunk gather without any process group: what every rank writes into its slot of chunk c's [world, width] block,
    followed by `_place`, must put prompt i's row at row i of the result -- for equal shards (one strided copy) and ragged ones (a copy per rank)."""
    bounds = [cdist.shard_range(total, r, world) for r in range(world)]
    sizes = [h - l for l, h in bounds]
    n_chunks = max(1, -(-max(sizes) // chunk))
    out = torch.full((total, 2), -1.0)
    for c in range(n_chunks):
        offs = [min(l + c * chunk, h) for l, h in bounds]
        cnts = [min(l + (c + 1) * chunk, h) - o for (l, h), o in zip(bounds, offs)]
        width = max(cnts)
        assert 0 < width <= min(chunk, max(sizes))
        block = torch.full((world, width, 2), float("nan"))          # the surplus rows of a ragged block hold garbage: never copied out
        for r in range(world):                                        # (what the all-gather leaves on every rank)
            block[r, : cnts[r]] = torch.arange(offs[r], offs[r] + cnts[r], dtype=torch.float32).view(-1, 1).expand(-1, 2)
        cdist._place(out, block, offs, cnts, sizes, c * chunk)
    assert torch.equal(out, torch.arange(total, dtype=torch.float32).view(-1, 1).expand(-1, 2))


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
    cdist.reset_comm()
    feats, cov_rows, sum_fn, finish_fn = _flow_cpu_hooks()
    lo, hi = cdist.shard_range(S, rank, world)
    local = _flow_inputs(S)[..., lo:hi].contiguous()
    cov = cdist.sharded_flow_corrs(local, downsample=2, use_covariance=True, gather=True, features_fn=feats, cov_rows_fn=cov_rows)
    slab = cdist.sharded_flow_corrs(local, downsample=2, use_covariance=False, gather=False, features_fn=feats, cov_rows_fn=cov_rows)
    mm = cdist.sharded_mean_motion_map(local, normalize_per_sample=True, normalize=True, sum_fn=sum_fn, finish_fn=finish_fn)
    torch.save((cov, slab, mm), os.path.join(out_dir, "f%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,S", [(2, 12), (2, 7), (2, 1), (4, 7)])
def test_sharded_flow_statistics_match_single_process(tmp_path, world, S):
    from oracle import flowstats_oracle as FO

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
