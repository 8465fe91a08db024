"""oracle/flowstats_oracle.py against outputs of the reference itself (tests/golden/flowstats.npz, written by
tests/golden/make_golden.py::run_flowstats_case from cwm/models/segmentation.py:250-276, 479-547)."""
import os
import sys

import numpy as np
import torch

from oracle import flowstats_oracle as FO

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)


def _inputs():
    from make_golden import flowstats_inputs  # the seeded generator only; the reference is not imported here

    return flowstats_inputs()


def test_oracle_matches_reference_outputs():
    g = np.load(os.path.join(GOLDEN, "flowstats.npz"))
    fl = _inputs()
    assert tuple(g["shape"]) == tuple(fl.shape)
    for ds in (1, 2, 4):
        rows = slice(0, 2) if ds == 1 else slice(None)
        assert np.array_equal(FO.compute_flow_corrs(fl, ds, True).numpy()[:, :, rows], g["cov_ds%d" % ds])
        assert np.array_equal(FO.compute_flow_corrs(fl, ds, False).numpy()[:, :, rows], g["corr_ds%d" % ds])
    for nps in (0, 1):
        for nm in (0, 1):
            assert np.array_equal(FO.compute_mean_motion_map(fl, bool(nps), bool(nm)).numpy(), g["map_nps%d_n%d" % (nps, nm)])
    assert np.array_equal(FO.compute_flow_samples_magnitude(fl, True).numpy(), g["mag_norm"])
    assert np.array_equal(FO.compute_mean_motion_map(fl[..., 0].norm(dim=1, keepdim=True)).numpy(), g["map_4d"])
    # a single sample has no covariance: torch.cov gives NaN, the reference zeroes it (segmentation.py:541)
    one = FO.compute_flow_corrs(fl[..., :1], 2, True).numpy()
    assert np.array_equal(one, g["cov_one_sample"]) and not one.any()


def test_option_prologues_match_reference_outputs():
    """The optional prologues of compute_flow_corrs (segmentation.py:519-538: thresh / binarize / range_thresh / normalize / zscore /
    Spearman argsort, alone and combined) against the reference's own outputs for each (tests/golden/flowstats_options.npz)."""
    from make_golden import FLOW_OPTION_CASES

    g = np.load(os.path.join(GOLDEN, "flowstats_options.npz"))
    fl = _inputs()
    for name, kw in FLOW_OPTION_CASES.items():
        out = FO.compute_flow_corrs(fl, 2, **kw).numpy()
        assert out.shape == g[name].shape and np.array_equal(out, g[name]), name
    # the options change the result (the fixtures exercise them)
    base = FO.compute_flow_corrs(fl, 2, True).numpy()
    assert all(np.abs(g[k] - base).max() > 1e-3 for k in ("thresh_cov", "range_thresh_cov", "normalize_cov", "spearman_zscore_cov"))


def test_covariance_properties():
    """Size-independent properties used again at full size on the GPU: symmetry, non-negative diagonal, constant sample rows
    have zero variance, sharding the samples and adding the sufficient statistics reproduces the covariance."""
    fl = _inputs()
    c = FO.compute_flow_corrs(fl, 2, True)[0, 0].reshape(64, 64)
    assert torch.allclose(c, c.t(), atol=1e-6) and (c.diagonal() >= 0).all()
    x = FO.flow_features(fl, 2)[0].double()
    S = x.shape[1]
    a, b = x[:, :5], x[:, 5:]
    s1, s2 = a.sum(1) + b.sum(1), a @ a.t() + b @ b.t()
    merged = (s2 - torch.outer(s1, s1) / S) / (S - 1)
    assert torch.allclose(merged.float(), c, atol=1e-5)
