"""The drop-in boundary, pinned from the REFERENCE side: the reference's own wrapper classes (`FlowGenerator`,
prediction.py:17 / segmentation.py:62, imported from /root/reference in the build container) run over THIS package's
predictor modules with only the library call stubbed, and the seam `self.predictor(self._preprocess(x), mask, ...)`
(prediction.py:419-422) must receive what `cwm_forward` / `cwm_conj_forward` are declared to take -- shapes, dtypes, keyword
names -- and every attribute the wrapper reads must exist.  CPU only; skipped where the reference tree is absent (the GPU box)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import ref_import  # noqa: E402

from counterfactualworldmodels_amd import config as C, conjoined_vmae as CV, vmae  # noqa: E402
from test_conj_oracle import TINY_CONJ  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_import.reference_available(), reason="reference tree not present (build container only)")


class DummyFlow(torch.nn.Module):
    def forward(self, x, backward=False, **k):
        return torch.zeros(x.shape[0], x.shape[1] - 1, 2, *x.shape[-2:])


@pytest.fixture(scope="module")
def ns():
    return ref_import.import_reference()


def _stub_plain(model, calls):
    def forward(x, mask, *args, **kwargs):
        calls.append((x, mask, args, kwargs))
        B, Nt = mask.shape
        n_masked = int(mask[0].sum())
        assert (mask.sum(-1) == n_masked).all(), "rows must be rectangular at the seam (vmae.py:167)"
        return torch.zeros(B, n_masked if n_masked else Nt, model.cfg.out_dim)

    model.forward = forward


def test_reference_wrapper_over_plain_predictor(ns):
    cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
    model = vmae.base_8x8patch_2frames_1tube()
    calls = []
    _stub_plain(model, calls)
    Psi = ns.segmentation.FlowGenerator(predictor=model, flow_model=DummyFlow(), imagenet_normalize_inputs=True, temporal_dim=2, seed=0)
    assert Psi.patch_size == (1, 8, 8) and model.t_dim == 2 and model.c_dim == 1   # attributes written / read by the wrapper
    B = 2
    x = torch.rand(B, 2, 3, 224, 224)
    Psi.set_input(x)
    assert tuple(Psi.mask_shape) == (2, 28, 28) == tuple(model.mask_size)
    mask = Psi.get_zeros_mask().clone()
    mask[:, 784 + 5] = False
    y = Psi.predict(x, mask, frame=-1)
    xs, ms, args, kwargs = calls[-1]
    assert tuple(xs.shape) == (B, 3, 2, 224, 224) and xs.dtype == torch.float32   # [B,C,T,H,W], imagenet-normalised
    assert tuple(ms.shape) == (B, 1568) and ms.dtype == torch.bool and not args and not kwargs
    mean = torch.tensor(C.IMAGENET_MEAN).view(1, 3, 1, 1, 1)
    std = torch.tensor(C.IMAGENET_STD).view(1, 3, 1, 1, 1)
    assert torch.allclose(xs, (x.transpose(1, 2) - mean) / std, atol=1e-6)
    assert tuple(y.shape) == (B, 1, 3, 224, 224)
    assert model.image_size == (224, 224) or tuple(model.image_size) == (224, 224)
    # the batch driver: S prompts over one image -> '(b s)' rows at the seam, flows from the plugged flow model
    calls.clear()
    n = 784
    S = 3
    active = torch.ones(1, 2 * n, S, dtype=torch.bool)
    active[:, :n] = False
    for s in range(S):
        active[0, n + 10 + s, s] = False
    ys, fs = Psi.predict_counterfactual_videos_and_flows(x[:1, 0], active_patches=active, shifts=[[1, 0], [0, 1], [-1, 1]], num_samples=S,
                                                         sample_batch_size=64)
    assert tuple(ys.shape) == (S, 2, 3, 224, 224) and tuple(fs.shape) == (S, 1, 2, 224, 224)
    assert sum(c[0].shape[0] for c in calls) == S and all(c[1].shape[1] == 1568 and c[1].dtype == torch.bool for c in calls)
    assert all(int(c[1][0].sum()) == 783 for c in calls)   # frame 2 masked except the moved patch
    assert [tuple(int(v) for v in s_) for s_ in Psi.shifts] == [(1, 0), (0, 1), (-1, 1)]
    # checkpoint surface: reference key names / shapes (prediction.py:81-107)
    sd = model.state_dict()
    assert tuple(sd["encoder.patch_embed.proj.weight"].shape) == (768, 3, 1, 8, 8) and "encoder_to_decoder.weight" in sd and "mask_token" in sd
    assert sum(v.numel() for v in sd.values()) == 92661312


def test_reference_wrapper_over_conjoined_predictor(ns):
    cfg = TINY_CONJ
    model = CV.ConjoinedPaddedVisionTransformer(cfg)
    calls = []

    def forward(x, mask, timestamps=None, x_context=None, mask_context=None, output_main=None, output_context=None, *args, **kwargs):
        calls.append(dict(x=x, mask=mask, x_context=x_context, mask_context=mask_context, extra=(args, kwargs)))
        vis = (~mask).sum(-1)
        vmax = int(vis.max())
        model._record_padding_state(mask, vis, vmax)   # what the real forward leaves behind (conjoined_vmae.py:49-116)
        return torch.zeros(mask.shape[0], mask.shape[1] + cfg.main_max_pad - vmax, cfg.main.out_dim)

    model.forward = forward
    Psi = ns.segmentation.FlowGenerator(predictor=model, flow_model=DummyFlow(), imagenet_normalize_inputs=True, temporal_dim=2, seed=0)
    # attributes the wrapper / UI read on a conjoined predictor
    assert model.context_stream.encoder.num_tokens == cfg.ctx_tokens and model.context_stream.patch_size[0] == cfg.ctx_tubelet
    assert model.get_context_input.num_channels == 6 and model.main_stream.max_padding_tokens == cfg.main_max_pad
    assert not hasattr(model, "padding_mask")          # before a forward, as in the reference (conjoined_vmae.py:347-354)
    B, n = 2, cfg.main.tokens_per_frame
    x = torch.rand(B, 2, 3, 32, 32)
    mask = torch.zeros(B, 2 * n, dtype=torch.bool)
    mask[:, n:] = True
    mask[:, n + 3] = False
    imu = torch.randn(B, 6, cfg.ctx_seq_len) * 0.1
    mc = torch.zeros(B, cfg.ctx_tokens, dtype=torch.bool)
    y = Psi.predict(x, mask, frame=None, reset_masks=False, x_context=imu, mask_context=mc)
    c = calls[-1]
    assert tuple(c["x"].shape) == (B, 3, 2, 32, 32) and tuple(c["mask"].shape) == (B, 2 * n) and c["mask"].dtype == torch.bool
    assert c["x_context"] is imu and c["mask_context"] is mc and c["extra"] == ((), {})
    assert tuple(y.shape) == (B, 2, 3, 32, 32)         # pad rows dropped, `get_current_inputs` used for the un-embed
    assert hasattr(model, "padding_mask") and tuple(model.padding_mask.shape) == (B, cfg.main_max_pad)
    (cx, cm, _), = model.get_current_inputs(Psi._preprocess(x), mask, x_context=imu, mask_context=mc)
    assert tuple(cx.shape) == (B, 3, 2, 32, 32) and torch.equal(cm, mask)
    Psi.reset_padding_masks()
    assert not hasattr(model, "padding_mask")
