"""Multi-GPU sharding of the batched counterfactual-sampling loop (BASELINE configs[3]).

The S counterfactual prompts over one frame pair are independent forwards (the reference merely
chunks them: `prediction.py:513-540`, `segmentation.py:423-430`), so they shard across ranks with
no collective inside the path: one process per GPU, weights replicated.  Around the path there are
exactly two collectives on `torch.distributed` (backend "nccl" = RCCL over xGMI on GPUs; "gloo" in
the CPU tests):

  1. broadcast from rank 0: the frame pair and the prompt table (a few KB)
  2. all-gather of the per-rank predictions

`predict_fn(x[b,T,C,H,W], mask[b,Nt]) -> y[b,...]` is the HIP predictor in production
(`PredictorBasedGenerator.predict`); the tests pass a CPU stand-in so the sharding logic is covered
without a GPU.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `total` units owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def broadcast_inputs(x: Optional[torch.Tensor], prompts: Optional[torch.Tensor], device, src: int = 0):
    """Rank `src` holds the frame pair x[1,T,C,H,W] (float32) and the prompt table [S,K] (int32);
    every rank returns its own copies.  Shapes are sent first so non-src ranks need no metadata."""
    rank, world = _world()
    if world == 1:
        return x.to(device), prompts.to(device)
    meta = torch.zeros(8, dtype=torch.int64, device=device)
    if rank == src:
        meta[:5] = torch.tensor(x.shape, dtype=torch.int64)
        meta[5:7] = torch.tensor(prompts.shape, dtype=torch.int64)
    dist.broadcast(meta, src=src)
    xs, ps = [int(v) for v in meta[:5]], [int(v) for v in meta[5:7]]
    xb = x.to(device=device, dtype=torch.float32).contiguous() if rank == src else torch.empty(xs, dtype=torch.float32, device=device)
    pb = prompts.to(device=device, dtype=torch.int32).contiguous() if rank == src else torch.empty(ps, dtype=torch.int32, device=device)
    dist.broadcast(xb, src=src)
    dist.broadcast(pb, src=src)
    return xb, pb


def all_gather_rows(y_local: torch.Tensor, total_rows: int) -> torch.Tensor:
    """Concatenate per-rank row blocks (rank order = prompt order).  Rank slices may differ by one
    row, so blocks are padded to the largest slice for the collective and trimmed afterwards."""
    rank, world = _world()
    if world == 1:
        return y_local
    max_rows = (total_rows + world - 1) // world
    pad = torch.zeros((max_rows,) + tuple(y_local.shape[1:]), dtype=y_local.dtype, device=y_local.device)
    pad[: y_local.shape[0]] = y_local
    out = torch.empty((world * max_rows,) + tuple(y_local.shape[1:]), dtype=y_local.dtype, device=y_local.device)
    dist.all_gather(list(out.chunk(world, 0)), pad)
    pieces = []
    for r in range(world):
        lo, hi = shard_range(total_rows, r, world)
        pieces.append(out[r * max_rows : r * max_rows + (hi - lo)])
    return torch.cat(pieces, 0)


def sharded_counterfactual_predictions(
    x: Optional[torch.Tensor],
    prompts: Optional[torch.Tensor],
    build_fn: Callable[[torch.Tensor, torch.Tensor], Tuple[torch.Tensor, torch.Tensor]],
    predict_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
    device,
    chunk: int = 32,
    gather: bool = True,
) -> torch.Tensor:
    """The config-4 loop: S prompts over ONE frame pair, sharded over the ranks.

    x, prompts: valid on rank 0 (ignored elsewhere).  `build_fn(x, prompts_slice)` turns a slice of
    the prompt table into per-prompt inputs (x_s[b,T,C,H,W], mask_s[b,Nt]) on `device`;
    `predict_fn` runs the predictor on at most `chunk` prompts at a time (the reference's
    `sample_batch_size`).  Returns all S predictions in prompt order on every rank (or only the
    local slice with gather=False)."""
    rank, world = _world()
    xb, pb = broadcast_inputs(x, prompts, device)
    S = pb.shape[0]
    lo, hi = shard_range(S, rank, world)
    outs = []
    for c0 in range(lo, hi, chunk):
        c1 = min(c0 + chunk, hi)
        xs, ms = build_fn(xb, pb[c0:c1])
        outs.append(predict_fn(xs, ms))
    if outs:
        y_local = torch.cat(outs, 0)
    else:  # a rank with an empty slice still has to join the collective with the right trailing shape
        xs, ms = build_fn(xb, pb[:1])
        y_local = predict_fn(xs, ms)[:0]
    return all_gather_rows(y_local, S) if gather else y_local


# ---- sample statistics over sharded samples -----------------------------------------------------------
def all_gather_last_axis(x_local: torch.Tensor) -> torch.Tensor:
    """Concatenate per-rank tensors along the LAST axis (rank order = sample order); sizes may differ (or be 0)."""
    rank, world = _world()
    if world == 1:
        return x_local
    n = torch.tensor([x_local.shape[-1]], dtype=torch.int64, device=x_local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    width = max(max(counts), 1)
    pad = torch.zeros(tuple(x_local.shape[:-1]) + (width,), dtype=x_local.dtype, device=x_local.device)
    pad[..., : x_local.shape[-1]] = x_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad.contiguous())
    return torch.cat([p[..., :c] for p, c in zip(parts, counts)], -1)


def sharded_flow_corrs(flow_samples_local: torch.Tensor, downsample: int = 1, use_covariance: bool = True, gather: bool = False,
                       features_fn: Optional[Callable] = None, cov_rows_fn: Optional[Callable] = None) -> torch.Tensor:
    """`FlowGenerator.compute_flow_corrs` (segmentation.py:479-547) when the S flow samples are spread over the ranks
    (`flow_samples_local` [B,C,H,W,S_rank]).  One collective on small data: the pooled features [B,P,S_rank] are all-gathered
    (P*S*4 bytes per frame pair: 12.8 MB at P = 12544, S = 256); then every rank computes rows `shard_range(P, rank, world)`
    of the [P,P] matrix -- the 629-MB result stays sharded unless `gather`.  Returns [B,nrows,P] ([B,P,P] with gather).
    `features_fn` / `cov_rows_fn` default to the HIP kernels (flowstats.py); the CPU tests inject the oracle."""
    if features_fn is None or cov_rows_fn is None:
        from . import flowstats

        features_fn = features_fn or flowstats.flow_features
        cov_rows_fn = cov_rows_fn or flowstats.feature_cov_rows
    rank, world = _world()
    B, _, H, W, S_local = flow_samples_local.shape
    ds = int(downsample or 1)
    if S_local > 0:
        x_local = features_fn(flow_samples_local, ds)
    else:  # a rank with an empty shard still joins the collective
        x_local = torch.zeros((B, (H // ds) * (W // ds), 0), dtype=torch.float32, device=flow_samples_local.device)
    x = all_gather_last_axis(x_local)
    P = x.shape[1]
    lo, hi = shard_range(P, rank, world)
    slab = cov_rows_fn(x, lo, hi - lo, use_covariance)
    if not gather or world == 1:
        return slab
    return all_gather_rows(slab.transpose(0, 1).contiguous(), P).transpose(0, 1).contiguous()


def sharded_mean_motion_map(flows_local: torch.Tensor, normalize_per_sample: bool = False, normalize: bool = True, eps: float = 1e-2,
                            sum_fn: Optional[Callable] = None, finish_fn: Optional[Callable] = None) -> torch.Tensor:
    """`FlowGenerator.compute_mean_motion_map` (segmentation.py:257-276) over sharded samples: per-rank sums of the (per-sample
    normalised) magnitudes, ONE all-reduce of [B,1,H,W] (+ the sample count), then the range normalisation on every rank."""
    if sum_fn is None or finish_fn is None:
        from . import flowstats

        sum_fn = sum_fn or flowstats.motion_map_sum
        finish_fn = finish_fn or flowstats.finish_motion_map
    rank, world = _world()
    B, _, H, W, S = flows_local.shape
    total = sum_fn(flows_local, normalize_per_sample, eps) if S > 0 else torch.zeros((B, 1, H, W), dtype=torch.float32, device=flows_local.device)
    n = torch.tensor([S], dtype=torch.int64, device=flows_local.device)
    if world > 1:
        dist.all_reduce(total)
        dist.all_reduce(n)
    return finish_fn(total, int(n.item()), normalize, eps)
