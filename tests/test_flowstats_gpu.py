"""Flow-sample statistics on a real MI355X, through the host mirror + C ABI: against the reference's own
outputs (tests/golden/flowstats.npz), against the CPU oracle on other layouts, and size-independent properties at the full
112 x 112 grid (a 12544 x 12544 matrix per frame pair)."""
import os
import sys

import numpy as np
import pytest
import torch

from counterfactualworldmodels_amd import flowstats as FS
from oracle import flowstats_oracle as FO

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
TOL = 1e-5  # fp32 statistics; the reference's torch.cov sums in a different order


def _golden_inputs():
    from make_golden import flowstats_inputs

    return flowstats_inputs()


def test_matches_reference_outputs():
    g = np.load(os.path.join(GOLDEN, "flowstats.npz"))
    fl = _golden_inputs().cuda()
    for ds in (1, 2, 4):
        rows = slice(0, 2) if ds == 1 else slice(None)
        cov = FS.compute_flow_cov(fl, downsample=ds).cpu().numpy()[:, :, rows]
        corr = FS.compute_flow_corrs(fl, downsample=ds).cpu().numpy()[:, :, rows]
        assert cov.shape == g["cov_ds%d" % ds].shape
        assert np.abs(cov - g["cov_ds%d" % ds]).max() <= TOL * max(1.0, np.abs(g["cov_ds%d" % ds]).max())
        assert np.abs(corr - g["corr_ds%d" % ds]).max() <= 1e-5
    for nps in (0, 1):
        for nm in (0, 1):
            m = FS.compute_mean_motion_map(fl, normalize_per_sample=bool(nps), normalize=bool(nm)).cpu().numpy()
            ref = g["map_nps%d_n%d" % (nps, nm)]
            assert m.shape == ref.shape and np.abs(m - ref).max() <= TOL * max(1.0, np.abs(ref).max())
    m4 = FS.compute_mean_motion_map(fl[..., 0].norm(dim=1, keepdim=True)).cpu().numpy()
    assert np.abs(m4 - g["map_4d"]).max() <= TOL
    assert np.abs(FS.compute_flow_samples_magnitude(fl).cpu().numpy() - g["mag_norm"]).max() <= TOL
    one = FS.compute_flow_cov(fl[..., :1], downsample=2).cpu().numpy()
    assert one.shape == g["cov_one_sample"].shape and not one.any()   # NaN -> 0 (segmentation.py:541)


def test_sample_major_layout_and_ragged_sizes_match_oracle():
    """The flow model emits [(b s),C,H,W]; the reference permutes it to [B,C,H,W,S] (segmentation.py:242-243): a strided view, read
    in place.  Sizes that are not multiples of the 128 x 128 tile or of the 32-sample staging step."""
    g = torch.Generator().manual_seed(3)
    B, S, H, W = 2, 37, 24, 40
    raw = torch.randn(B * S, 2, H, W, generator=g)
    view = raw.view(B, S, 2, H, W).permute(0, 2, 3, 4, 1)   # [B,C,H,W,S], S stride = C*H*W
    dev = raw.cuda().view(B, S, 2, H, W).permute(0, 2, 3, 4, 1)
    assert not dev.is_contiguous()
    for ds, cov in ((1, True), (2, False), (4, True)):
        out = FS.compute_flow_corrs(dev, downsample=ds, use_covariance=cov).cpu()
        ref = FO.compute_flow_corrs(view.contiguous(), ds, cov)
        assert out.shape == ref.shape and (out - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item()), (ds, cov)
    m = FS.compute_mean_motion_map(dev, normalize_per_sample=True).cpu()
    assert (m - FO.compute_mean_motion_map(view.contiguous(), normalize_per_sample=True)).abs().max().item() <= TOL
    # top-k samples and the "swap" concatenation (segmentation.py:500-511)
    out = FS.compute_flow_corrs(dev, flow_samples_swap=dev.flip(-1), downsample=4, take_top_k=10, use_covariance=True).cpu()
    x = torch.cat([FO.flow_features(view[..., :10].contiguous(), 4), FO.flow_features(view.flip(-1)[..., :10].contiguous(), 4)], -1)
    ref = torch.stack([torch.cov(x[b]) for b in range(B)]).view(out.shape)
    assert (out - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("S,H,W", [(24, 28, 36), (64, 20, 28), (128, 12, 20), (256, 16, 12), (512, 8, 12), (100, 10, 14), (1, 9, 7), (700, 6, 6)])
def test_motion_map_kernel_forms_match_oracle(S, H, W):
    """`compute_mean_motion_map` picks its kernels by layout and sample count: the register form (S = 64 / 128: 16 / 32 lanes per pixel; S a multiple of 256: whole-wave
    16-byte loads), the LDS-tile form (any other S of the packed [B, C, H, W, S] layout), the strided form (sample-major views, rows too long for the tile).  Every
    form, with and without the per-sample range normalisation, against the oracle's restatement of segmentation.py:250-276; pixel counts that do not fill the last
    workgroup; two movies; three channels."""
    g = torch.Generator().manual_seed(100 + S)
    for C_ in (2, 3):
        fl = torch.randn(2, C_, H, W, S, generator=g) * 1.5
        dev = fl.cuda()
        for nps in (False, True):
            for nm in (False, True):
                out = FS.compute_mean_motion_map(dev, normalize_per_sample=nps, normalize=nm).cpu()
                ref = FO.compute_mean_motion_map(fl, normalize_per_sample=nps, normalize=nm)
                tol = max(TOL, 4e-8 * S)   # (an fp32 sum over S samples, in whichever order the kernel form adds them)
                assert out.shape == ref.shape and (out - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item()), (S, C_, nps, nm)
        # the same samples as a sample-major strided view (the flow model's output order): the strided kernels
        sm = fl.permute(0, 4, 1, 2, 3).contiguous().cuda().permute(0, 2, 3, 4, 1)
        assert S == 1 or not sm.is_contiguous()
        out = FS.compute_mean_motion_map(sm, normalize_per_sample=True).cpu()
        assert (out - FO.compute_mean_motion_map(fl, normalize_per_sample=True)).abs().max().item() <= max(TOL, 4e-8 * S)


@pytest.mark.parametrize("S", [8, 24, 37, 256])
def test_pooled_features_vector_and_scalar_forms(S):
    """`flow_features` (segmentation.py:503-513): the 16-byte form (sample axis innermost, S a multiple of 4) and the strided scalar form against the oracle, for every
    pooling factor the interface uses; a sample-major view of the same samples takes the scalar form and must agree with the packed layout."""
    g = torch.Generator().manual_seed(S)
    fl = torch.randn(2, 2, 16, 24, S, generator=g)
    sm = fl.permute(0, 4, 1, 2, 3).contiguous().cuda().permute(0, 2, 3, 4, 1)
    for ds in (1, 2, 4):
        out = FS.flow_features(fl.cuda(), ds)
        ref = FO.flow_features(fl, ds)
        assert out.shape == ref.shape and (out.cpu() - ref).abs().max().item() <= 1e-6
        assert (FS.flow_features(sm, ds) - out).abs().max().item() <= 1e-6


@pytest.mark.parametrize("P,S", [(12544, 256), (12544, 24), (300, 70), (113, 5), (1, 8)])
def test_column_statistics_at_full_size(P, S):
    """The z-score / normalise / range prologues reduce over the P positions of every sample column: chunked one-pass Welford in float64, merged in a fixed order
    (flowstats.hip flow_colpartial_kernel).  Against torch in float64 at the bench size (P = 12544: 112 chunks), at sizes that end inside a chunk, and run to run."""
    g = torch.Generator().manual_seed(P + S)
    x = (torch.randn(1, P, S, generator=g) * 3 + 1.5).abs()
    xd = x.cuda()
    z = FS.transform_features(xd.clone(), zscore=True)
    assert torch.equal(torch.nan_to_num(z), torch.nan_to_num(FS.transform_features(xd.clone(), zscore=True)))      # deterministic
    x64 = x[0].double()
    if P > 1:
        ref = ((x64 - x64.mean(0)[None]) / x64.std(0).clamp(min=1e-12)[None]).float()
        assert (z[0].cpu() - ref).abs().max().item() <= 2e-5
    else:
        assert torch.isnan(z).all()
    n = FS.transform_features(xd.clone(), normalize=True)[0].cpu()
    assert (n - x[0] / x[0].amax(0, True).clamp(min=1e-12)).abs().max().item() <= 1e-6
    r = FS.transform_features(xd.clone(), range_thresh=0.3)[0].cpu()
    sh = x[0] - x[0].amin(0, True)
    assert torch.equal(r, (sh > 0.3 * sh.amax(0, True)).float())


@pytest.mark.parametrize("P,S", [(200, 33), (129, 4), (384, 256), (50, 2), (130, 100)])
def test_symmetric_covariance_tiles_match_torch(P, S):
    """The whole-matrix call computes the 128 x 128 tiles on and above the diagonal and writes every off-diagonal tile twice (once mirrored): against torch.cov /
    torch.corrcoef in float64 for sizes with a ragged last tile, one tile, three tile rows, S below / not a multiple of the 32-sample step -- and the row-slab
    (rectangular-grid) kernel must give the same bits for any slab."""
    g = torch.Generator().manual_seed(P * 7 + S)
    x = torch.randn(2, P, S, generator=g).cuda()
    for use_cov in (True, False):
        full = FS.feature_cov_rows(x, 0, P, use_cov)
        for b in range(2):
            x64 = x[b].double().cpu()
            ref = (torch.cov(x64) if use_cov else torch.corrcoef(x64)).float()
            assert (full[b].cpu() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item()), (P, S, use_cov)
        assert torch.equal(full, full.transpose(1, 2))
        for row0, nrows in ((0, min(P, 7)), (P // 3, P - P // 3), (P - 1, 1)):
            assert torch.equal(FS.feature_cov_rows(x, row0, nrows, use_cov), full[:, row0:row0 + nrows])


def test_option_prologues_match_reference_outputs():
    """compute_flow_corrs with its optional prologues on the device (segmentation.py:519-538; `cwm_flow_transform`): thresh, binarize,
    range_thresh, normalize, zscore and the Spearman argsort, alone and combined, against the reference's own outputs; a strided
    sample-major input and the row-slab form go through the same prologue."""
    from make_golden import FLOW_OPTION_CASES

    g = np.load(os.path.join(GOLDEN, "flowstats_options.npz"))
    fl = _golden_inputs().cuda()
    for name, kw in FLOW_OPTION_CASES.items():
        out = FS.compute_flow_corrs(fl, downsample=2, **kw).cpu().numpy()
        ref = g[name]
        err = np.abs(out - ref).max()
        assert out.shape == ref.shape and err <= TOL * max(1.0, np.abs(ref).max()), (name, err)
    # binary / rank features are exact small integers: the statistics of those cases agree to the last bits of the fp32 sums
    x = FS.flow_features(fl, 2)
    xb = FS.transform_features(x.clone(), thresh=1.5, binarize=True)
    assert set(torch.unique(xb).tolist()) <= {0.0, 1.0} and torch.equal(xb, (x > 1.5).float())
    xr = FS.transform_features(x.clone(), do_spearman=True)
    assert torch.equal(xr.cpu(), torch.argsort(x.cpu(), dim=-1, stable=True).float())
    # ragged sizes (S not a multiple of 64, P not of 4), strided input, row slab
    gen = torch.Generator().manual_seed(5)
    raw = torch.randn(2 * 37, 2, 12, 20, generator=gen)
    dev = raw.cuda().view(2, 37, 2, 12, 20).permute(0, 2, 3, 4, 1)
    view = raw.view(2, 37, 2, 12, 20).permute(0, 2, 3, 4, 1).contiguous()
    for kw in (dict(range_thresh=0.3, normalize=True), dict(zscore=True, use_covariance=True), dict(do_spearman=True, use_covariance=True)):
        out = FS.compute_flow_corrs(dev, downsample=4, **kw).cpu()
        ref = FO.compute_flow_corrs(view, 4, **kw)
        assert (out - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item()), kw
    slab = FS.compute_flow_corrs(dev, downsample=4, zscore=True, use_covariance=True, rows=(3, 7)).cpu()
    full = FO.compute_flow_corrs(view, 4, zscore=True, use_covariance=True).reshape(2, 15, 15)
    assert (slab - full[:, 3:10]).abs().max().item() <= TOL * max(1.0, full.abs().max().item())


def test_user_distance_func_and_nan_columns():
    """`compute_flow_corrs(distance_func=<callable>)` (segmentation.py:485, :503-513): the caller's function runs as PyTorch on the device over the
    pooled samples, its [B, P, S] result goes through the device prologues / covariance -- against the oracle's restatement with the same callable
    (the reference's own default passed explicitly must reproduce the fused default path).  And the NaN semantics of the prologues as torch has
    them: argsort orders NaNs last (always a permutation), amax / clamp / std propagate a NaN column, one position (P = 1) z-scores to NaN, and
    every NaN of the matrix becomes 0 (segmentation.py:541)."""
    g = torch.Generator().manual_seed(21)
    fl = torch.randn(2, 2, 16, 24, 12, generator=g)
    dev = fl.cuda()

    def l1(a, b):
        return (a - b).abs().mean(1, True)

    def channel_mse(a, b):   # utils.ChannelMSE(dim=1) restated: sqrt(mean_c((a - b)^2))
        return (a - b).square().mean(1, True).sqrt()

    for fn, kw in ((l1, dict(use_covariance=True)), (l1, dict(zscore=True, downsample=2)), (channel_mse, dict(use_covariance=True, downsample=2))):
        out = FS.compute_flow_corrs(dev, distance_func=fn, **kw).cpu()
        ds = kw.get("downsample", 1)
        pooled = torch.nn.functional.avg_pool3d(fl.permute(0, 1, 4, 2, 3), (1, ds, ds), stride=(1, ds, ds)).permute(0, 1, 3, 4, 2)
        x = fn(pooled, torch.zeros_like(pooled)).reshape(2, -1, 12)
        ref = []
        for b in range(2):
            xb = x[b]
            if kw.get("zscore"):
                xb = (xb - xb.mean(0)[None]) / xb.std(0).clamp(min=1e-12)[None]
            c = torch.cov(xb) if kw.get("use_covariance") else torch.corrcoef(xb)
            c[torch.isnan(c)] = 0
            ref.append(c)
        ref = torch.stack(ref).view(out.shape)
        assert (out - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item()), (fn.__name__, kw)
    same = FS.compute_flow_corrs(dev, distance_func=channel_mse, use_covariance=True, downsample=2)
    assert (same - FS.compute_flow_corrs(dev, use_covariance=True, downsample=2)).abs().max().item() <= TOL
    # NaN semantics
    x = torch.randn(1, 6, 9, generator=g)
    x[0, 2, 4] = float("nan")
    x[0, 2, 7] = float("nan")
    x[0, 5, 0] = float("nan")
    xr = FS.transform_features(x.cuda().clone(), do_spearman=True).cpu()
    assert torch.equal(xr, torch.argsort(x, dim=-1, stable=True).float())          # NaNs last, in index order: a permutation in every row
    xn = FS.transform_features(x.cuda().clone(), normalize=True).cpu()
    ref = x[0] / x[0].amax(0, True).clamp(min=1e-12)
    assert torch.equal(torch.isnan(xn[0]), torch.isnan(ref)) and (torch.nan_to_num(xn[0]) - torch.nan_to_num(ref)).abs().max().item() <= 1e-6
    xz = FS.transform_features(x.cuda().clone(), zscore=True).cpu()
    ref = (x[0] - x[0].mean(0)[None]) / x[0].std(0).clamp(min=1e-12)[None]
    assert torch.equal(torch.isnan(xz[0]), torch.isnan(ref)) and (torch.nan_to_num(xz[0]) - torch.nan_to_num(ref)).abs().max().item() <= 1e-5
    # a NaN-free column holding +inf AND -inf: its mean is NaN, but amax is +inf (x / inf = 0, inf / inf = NaN) -- the NaN is tracked, not read off the mean
    xi = torch.randn(1, 6, 9, generator=g)
    xi[0, 1, 3], xi[0, 4, 3] = float("inf"), float("-inf")
    xin = FS.transform_features(xi.cuda().clone(), normalize=True).cpu()
    ref = xi[0] / xi[0].amax(0, True).clamp(min=1e-12)
    assert torch.equal(torch.isnan(xin[0]), torch.isnan(ref)) and torch.equal(torch.nan_to_num(xin[0]), torch.nan_to_num(ref))
    assert not torch.isnan(xin[0, 0, 3]) and xin[0, 0, 3] == 0 and torch.isnan(xin[0, 1, 3])
    xir = FS.transform_features(xi.cuda().clone(), range_thresh=0.5).cpu()   # segmentation.py:529-532 on that column: nothing exceeds an infinite / NaN range
    sh = xi[0] - xi[0].amin(0, True)
    assert torch.equal(xir[0], (sh > 0.5 * sh.amax(0, True)).float())
    one = FS.transform_features(torch.randn(1, 1, 5, generator=g).cuda(), zscore=True)                                # P = 1: std = NaN
    assert torch.isnan(one).all()
    c = FS.compute_flow_corrs(torch.randn(1, 2, 1, 1, 5, generator=g).cuda(), zscore=True, use_covariance=True)      # ... and the matrix is 0
    assert c.shape == (1, 1, 1, 1, 1, 1) and c.item() == 0.0


def test_unsupported_inputs_fail_loudly():
    with pytest.raises(RuntimeError):
        FS.compute_flow_corrs(torch.zeros(1, 2, 8, 8, 4))   # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        FS.compute_flow_corrs(torch.zeros(1, 2, 8, 8, 4), distance_func=lambda a, b: (a - b).abs().mean(1, True))


def test_full_grid_properties():
    """224 x 224 flows, downsample 2 (the interface's setting): P = 12544, 629 MB of covariance.  Row slabs tile the matrix; it is
    symmetric; its diagonal is the per-position sample variance; a constant sample set has zero covariance."""
    g = torch.Generator().manual_seed(11)
    S = 24
    fl = (torch.randn(1, 2, 224, 224, S, generator=g) * 2).cuda()
    x = FS.flow_features(fl, 2)
    P = x.shape[1]
    assert P == 12544
    full = FS.feature_cov_rows(x, 0, P, True)[0]
    assert (full - full.t()).abs().max().item() <= 1e-5
    var = x[0].var(dim=1, unbiased=True)
    assert (full.diagonal() - var).abs().max().item() <= 1e-5
    for row0, nrows in ((0, 1), (3000, 1137), (P - 100, 100)):
        slab = FS.feature_cov_rows(x, row0, nrows, True)[0]
        assert torch.equal(slab, full[row0:row0 + nrows])
    sub = torch.tensor([0, 17, 5000, P - 1], device="cuda")
    ref = torch.cov(x[0][sub].double()).float()
    assert (full[sub][:, sub] - ref).abs().max().item() <= 1e-5
    corr = FS.feature_cov_rows(x, 100, 64, False)[0]
    assert corr.abs().max().item() <= 1.0 and (corr[torch.arange(64), torch.arange(100, 164)] - 1).abs().max().item() <= 1e-5
    const = torch.full((1, 2, 224, 224, 5), 0.5, device="cuda")
    assert not FS.compute_flow_cov(const, downsample=2).any()
    mm = FS.compute_mean_motion_map(fl)
    assert mm.shape == (1, 1, 224, 224) and mm.min().item() == 0.0 and abs(mm.max().item() - 1.0) <= 1e-6
