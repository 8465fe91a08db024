"""CPU restatement of the reference's flow-sample statistics.  TEST INFRASTRUCTURE ONLY: imported by
tests/, by `__graft_entry__.smoke()` and by nothing in the product path.

Follows cwm/models/segmentation.py:
  * `FlowGenerator.compute_flow_samples_magnitude`  :250-255
  * `FlowGenerator.compute_mean_motion_map`         :257-276
  * `FlowGenerator.compute_flow_corrs`              :479-547  (distance_func = utils.ChannelMSE(dim=1): cwm/models/utils.py:510-521)
Pinned: tests/golden/flowstats.npz holds outputs of the reference itself (tests/golden/make_golden.py::run_flowstats_case) for
seeded random flows; tests/test_flowstats_oracle.py checks this file against them.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def compute_flow_samples_magnitude(flows: torch.Tensor, normalize: bool = True, dim: int = -4, eps: float = 1e-2) -> torch.Tensor:
    """flows [..., C, H, W, S] -> [..., 1, H, W, S] (segmentation.py:250-255)."""
    mags = flows.square().sum(dim, True).sqrt().to(flows.dtype)
    if normalize:
        mags = mags - mags.amin((-3, -2), True)
        mags = mags / mags.amax((-3, -2), True).clamp(min=eps)
    return mags


def compute_mean_motion_map(flows: torch.Tensor, normalize_per_sample: bool = False, normalize: bool = True, dim: int = -4,
                            eps: float = 1e-2) -> torch.Tensor:
    """flows [B,C,H,W,S] -> [B,1,H,W]; a 4-D input is only range-normalised (segmentation.py:257-276)."""
    if flows.dim() == 5:
        motion_map = compute_flow_samples_magnitude(flows, normalize=normalize_per_sample, dim=dim, eps=eps).mean(-1)
    else:
        motion_map = flows
        normalize = True
    if normalize:
        motion_map = motion_map - motion_map.amin((-2, -1), True)
        motion_map = motion_map / motion_map.amax((-2, -1), True).clamp(min=eps)
    return motion_map


def flow_features(flow_samples: torch.Tensor, downsample: int = 1) -> torch.Tensor:
    """[B,C,H,W,S] -> [B, (H/ds)(W/ds), S]: ds x ds average pool per channel and sample, then the root of the channel-mean
    square (ChannelMSE against zeros) -- segmentation.py:503-513."""
    B, C, H, W, S = flow_samples.shape
    ds = downsample
    x = F.avg_pool3d(flow_samples.permute(0, 1, 4, 2, 3), (1, ds, ds), stride=(1, ds, ds)).permute(0, 1, 3, 4, 2)
    x = torch.sqrt(x.square().mean(1, True).float()).to(flow_samples.dtype)
    return x.reshape(B, -1, S)


def compute_flow_corrs(flow_samples: torch.Tensor, downsample: int = 1, use_covariance: bool = False, do_spearman: bool = False, thresh=None,
                       eps: float = 1e-12, binarize: bool = False, normalize: bool = False, zscore: bool = False, range_thresh=None) -> torch.Tensor:
    """[B,C,H,W,S] -> [B,1,H/ds,W/ds,H/ds,W/ds]: covariance (torch.cov, unbiased) or Pearson correlation (torch.corrcoef) of the
    pooled flow magnitude over the S samples, NaN -> 0 (segmentation.py:479-547), with the optional prologues of :519-538 on the
    [P, S] feature matrix of every batch element (statistics over dim 0 = the positions)."""
    B, C, H, W, S = flow_samples.shape
    if S == 0:
        flow_samples = torch.zeros(list(flow_samples.shape)[:-1] + [1], dtype=torch.float32)
    x = flow_features(flow_samples, downsample)
    out = []
    for b in range(B):
        xb = torch.argsort(x[b], -1).float() if do_spearman else x[b]                      # :520-523
        if thresh is not None and binarize is False:                                        # :525-526
            xb = xb * (xb > thresh).float()
        elif thresh is not None:                                                            # :527-528
            xb = (xb > thresh).float()
        elif range_thresh is not None:                                                      # :529-532
            xb = xb - xb.amin(0, True)
            xb = (xb > (range_thresh * xb.amax(0, True))).float()
        if normalize:                                                                       # :534-535
            xb = xb / xb.amax(0, True).clamp(min=eps)
        if zscore:                                                                          # :536-538
            mn, sd = xb.mean(0), xb.std(0).clamp(min=eps)
            xb = (xb - mn[None]) / sd[None]
        c = torch.cov(xb) if use_covariance else torch.corrcoef(xb)
        c[torch.isnan(c)] = 0
        out.append(c)
    ds = downsample
    return torch.stack(out, 0).view(B, 1, H // ds, W // ds, H // ds, W // ds)
