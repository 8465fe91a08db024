"""Flow-sample statistics on the device: host-side mirror of the reference functions that reduce the S counterfactual
flow samples.  Same names, argument meaning and output shapes as

    cwm/models/segmentation.py  FlowGenerator.compute_flow_samples_magnitude :250-255
                                FlowGenerator.compute_mean_motion_map        :257-276
                                FlowGenerator.compute_flow_corrs             :479-547
    cwm/interface.py            compute_flow_cov = partial(compute_flow_corrs, use_covariance=True)  :27-29

All arithmetic runs in `libcwm_hip.so` (flowstats.hip); there is no CPU fallback.  Options of `compute_flow_corrs` the demo
and interface never switch on (Spearman ranks, thresholds, z-scoring, a custom distance function) raise NotImplementedError
instead of silently computing something else.

Sharded form (samples spread over the ranks, as the prompt-sharding driver leaves them): `dist.sharded_flow_corrs` and
`dist.sharded_mean_motion_map` -- every rank computes the small feature matrix of its own samples, the features are
all-gathered, and each rank produces a slab of rows of the [P, P] matrix; motion-map sums are all-reduced before the final
normalisation.  They are built from `flow_features`, `feature_cov_rows`, `motion_map_sum` and `finish_motion_map` below.
"""
from __future__ import annotations

import ctypes as C
from functools import partial
from typing import Optional, Tuple

import torch

from . import _lib


def _strides5(flows: torch.Tensor):
    if flows.dim() != 5:
        raise RuntimeError("expected flow samples of shape [B,C,H,W,S], got %s" % (tuple(flows.shape),))
    if flows.dtype != torch.float32:
        flows = flows.float()
    return flows, (C.c_int64 * 5)(*flows.stride())


def _require_cuda(t: torch.Tensor, what: str):
    _lib.require_gpu()
    if not t.is_cuda:
        raise RuntimeError("%s needs a CUDA/HIP tensor (no CPU fallback); got %s" % (what, t.device))


def flow_features(flow_samples: torch.Tensor, downsample: int = 1) -> torch.Tensor:
    """[B,C,H,W,S] -> [B,(H/ds)(W/ds),S] pooled flow magnitudes (segmentation.py:503-513)."""
    _require_cuda(flow_samples, "flow_features")
    flows, strides = _strides5(flow_samples)
    B, Cc, H, W, S = flows.shape
    ds = int(downsample or 1)
    x = torch.empty((B, (H // ds) * (W // ds), S), device=flows.device, dtype=torch.float32)
    with torch.cuda.device(flows.device):
        _lib.check(_lib.get_lib().cwm_flow_features(flows.data_ptr(), strides, B, Cc, H, W, S, ds, x.data_ptr(),
                                                   _lib.current_stream_handle(flows.device)))
    return x


def feature_cov_rows(x: torch.Tensor, row0: int, nrows: int, use_covariance: bool) -> torch.Tensor:
    """Rows [row0, row0+nrows) of cov / corrcoef over the last axis of x [B,P,S] -> [B,nrows,P]."""
    _require_cuda(x, "feature_cov_rows")
    x = x.contiguous()
    B, P, S = x.shape
    out = torch.empty((B, nrows, P), device=x.device, dtype=torch.float32)
    xc = torch.empty_like(x)
    inv_std = torch.empty((B, P), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.get_lib().cwm_flow_cov(x.data_ptr(), B, P, S, int(row0), int(nrows), int(bool(use_covariance)), xc.data_ptr(),
                                              inv_std.data_ptr(), out.data_ptr(), _lib.current_stream_handle(x.device)))
    return out


def transform_features(x: torch.Tensor, do_spearman=False, thresh=None, binarize=False, normalize=False, zscore=False, range_thresh=None,
                       eps: float = 1e-12) -> torch.Tensor:
    """The optional prologues of `compute_flow_corrs` on the pooled features x [B,P,S], in the reference's order (segmentation.py:519-538):
    Spearman argsort over the samples of every position; `thresh` (x * (x > t), or (x > t) with `binarize`), else `range_thresh`
    (((x - min) > r * (max - min)) per sample column, min / max over the positions); `normalize` (x / max over positions); `zscore`
    ((x - mean) / std over positions).  In place on a contiguous copy; returns it."""
    if not (do_spearman or thresh is not None or range_thresh is not None or normalize or zscore):
        return x
    _require_cuda(x, "transform_features")
    x = x.contiguous()
    B, P, S = x.shape
    mode, t = 0, 0.0
    if thresh is not None:
        mode, t = (2 if binarize else 1), float(thresh)
    elif range_thresh is not None:
        mode, t = 3, float(range_thresh)
    work = None
    if mode == 3 or normalize or zscore:  # the column statistics + the partials of the chunked column pass
        work = torch.empty(int(_lib.get_lib().cwm_flow_transform_work_bytes(B, P, S)), device=x.device, dtype=torch.uint8)
    with torch.cuda.device(x.device):
        _lib.check(_lib.get_lib().cwm_flow_transform(x.data_ptr(), B, P, S, int(bool(do_spearman)), mode, t, int(bool(normalize)), int(bool(zscore)),
                                                    float(eps), _lib.ptr(work), _lib.current_stream_handle(x.device)))
    return x


def compute_flow_corrs(flow_samples, flow_samples_swap=None, downsample=1, take_top_k=None, do_spearman=False, distance_func=None,
                       thresh=None, use_covariance=False, eps=1e-12, binarize=False, normalize=False, zscore=False, range_thresh=None,
                       rows: Optional[Tuple[int, int]] = None):
    """[B,C,H,W,S] -> [B,1,H/ds,W/ds,H/ds,W/ds] covariance or correlation of the pooled flow magnitude over the samples
    (segmentation.py:479-547).  `rows=(row0, nrows)` returns only that slab [B,nrows,P] of the flattened [P,P] matrix."""
    _require_cuda(flow_samples, "compute_flow_corrs")
    B, Cc, H, W, S = flow_samples.shape
    if S == 0:  # segmentation.py:495-499
        flow_samples = torch.zeros(list(flow_samples.shape)[:-1] + [1], device=flow_samples.device, dtype=torch.float32)
        S = 1
    K = S if take_top_k is None else take_top_k
    ds = int(downsample or 1)
    if flow_samples_swap is not None:
        assert list(flow_samples_swap.shape) == [B, Cc, H, W, S]
    if distance_func is None:
        # the reference's default features, utils.ChannelMSE(dim=1) against zeros (segmentation.py:485, :513), fused with the pooling
        x = flow_features(flow_samples[..., :K], downsample)
        if flow_samples_swap is not None:
            x = torch.cat([x, flow_features(flow_samples_swap[..., :K], downsample)], -1)
    else:
        # a caller's `distance_func(flow_inp, zeros)` (segmentation.py:503-513): it is the caller's PyTorch code, so it runs as PyTorch -- on the
        # flows' device, on the pooled samples exactly as the reference hands them over -- and its [B, P, S] result goes through the same
        # device prologues and covariance kernels as the default features
        def _ds(fs):
            return torch.nn.functional.avg_pool3d(fs[..., :K].permute(0, 1, 4, 2, 3), (1, ds, ds), stride=(1, ds, ds)).permute(0, 1, 3, 4, 2)

        flow_inp = _ds(flow_samples.float())
        if flow_samples_swap is not None:
            flow_inp = torch.cat([flow_inp, _ds(flow_samples_swap.float())], -1)
        x = distance_func(flow_inp, torch.zeros_like(flow_inp)).reshape(B, -1, flow_inp.size(-1)).float().contiguous()
    P = x.shape[1]
    x = transform_features(x, do_spearman=do_spearman, thresh=thresh, binarize=binarize, normalize=normalize, zscore=zscore,
                           range_thresh=range_thresh, eps=eps)
    if rows is not None:
        return feature_cov_rows(x, rows[0], rows[1], use_covariance)
    return feature_cov_rows(x, 0, P, use_covariance).view(B, 1, H // ds, W // ds, H // ds, W // ds)


compute_flow_cov = partial(compute_flow_corrs, use_covariance=True)  # cwm/interface.py:27-29


def compute_flow_samples_magnitude(flows, normalize=True, dim=-4, eps=1e-2):
    """[...,C,H,W,S] -> [...,1,H,W,S] (segmentation.py:250-255).  Elementwise + one range per (b, s): plain torch ops on the
    caller's device (the fused reduction lives in compute_mean_motion_map)."""
    mags = flows.square().sum(dim, True).sqrt().to(flows.dtype)
    if normalize:
        mags = mags - mags.amin((-3, -2), True)
        mags = mags / mags.amax((-3, -2), True).clamp(min=eps)
    return mags


def motion_map_sum(flows: torch.Tensor, normalize_per_sample: bool = False, eps: float = 1e-2) -> torch.Tensor:
    """[B,C,H,W,S] -> [B,1,H,W] sum over the samples of the (optionally per-sample range-normalised) flow magnitude."""
    _require_cuda(flows, "motion_map_sum")
    flows, strides = _strides5(flows)
    B, Cc, H, W, S = flows.shape
    out = torch.empty((B, 1, H, W), device=flows.device, dtype=torch.float32)
    mm = torch.empty(int(_lib.get_lib().cwm_flow_motion_work_bytes(B, S)), device=flows.device, dtype=torch.uint8) if normalize_per_sample else None
    with torch.cuda.device(flows.device):
        _lib.check(_lib.get_lib().cwm_flow_motion_sum(flows.data_ptr(), strides, B, Cc, H, W, S, int(bool(normalize_per_sample)), float(eps),
                                                     _lib.ptr(mm), out.data_ptr(), _lib.current_stream_handle(flows.device)))
    return out


def finish_motion_map(map_sum: torch.Tensor, num_samples: int, normalize: bool = True, eps: float = 1e-2) -> torch.Tensor:
    _require_cuda(map_sum, "finish_motion_map")
    m = map_sum.contiguous().clone()
    B = m.shape[0]
    with torch.cuda.device(m.device):
        _lib.check(_lib.get_lib().cwm_flow_map_finish(m.data_ptr(), B, m[0].numel(), 1.0 / float(num_samples), int(bool(normalize)), float(eps),
                                                     _lib.current_stream_handle(m.device)))
    return m


def compute_mean_motion_map(flows, normalize_per_sample=False, normalize=True, dim=-4, eps=1e-2):
    """[B,C,H,W,S] -> [B,1,H,W]: mean over the samples of the flow magnitude, range-normalised over (H,W); a 4-D input is only
    range-normalised (segmentation.py:257-276)."""
    if flows.dim() == 5:
        if dim not in (-4, 1):
            raise NotImplementedError("compute_mean_motion_map: the channel axis must be dim -4")
        return finish_motion_map(motion_map_sum(flows, normalize_per_sample, eps), flows.shape[-1], normalize, eps)
    m = flows.float()
    return finish_motion_map(m.reshape(m.shape[0], 1, *m.shape[-2:]), 1, True, eps).view(m.shape)
