"""f-1 / f-2: device-side motion-counterfactual prompt construction against the fixture captured
from the reference's own `FlowGenerator.create_motion_counterfactuals` (bit-exact), and the batched
counterfactual prediction driver."""
import os

import numpy as np
import pytest
import torch

from counterfactualworldmodels_amd import config as C, prediction, synthetic as S, vmae
from oracle import vmae_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TINY = C.VmaeConfig(name="tiny_8x8", img_size=(32, 32), patch=8, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128,
                    dec_depth=1, dec_heads=2)
TINY_SPEC = O.VmaeSpec(img_size=(32, 32), patch=8, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128, dec_depth=1, dec_heads=2)


def wrapper(cfg, seed=3):
    m = vmae.PretrainVisionTransformer(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, seed).items()})
    return prediction.PredictorBasedGenerator(predictor=m.cuda().eval(), imagenet_normalize_inputs=True, temporal_dim=2)


@pytest.mark.parametrize("tag", ["tiny", "base8"])
def test_shift_prompts_bit_exact_vs_reference(tag):
    g = np.load(os.path.join(GOLDEN, "shift_prompts.npz"))
    cfg = TINY if tag == "tiny" else C.CONFIGS["base_8x8patch_2frames_1tube"]
    G = wrapper(cfg)
    x = torch.from_numpy(S.synthetic_frames(1, cfg, 21)).cuda()
    active = torch.from_numpy(g[f"{tag}_active"]).cuda()
    passive = torch.from_numpy(g[f"{tag}_passive"]).cuda()
    shifts = [tuple(int(v) for v in r) for r in g[f"{tag}_shifts"]]
    torch.manual_seed(int(g[f"{tag}_rect_seed"]))
    xs, ms = G.create_motion_counterfactuals(x, masks=passive, active_patches=active, shifts=shifts, fix_passive=True, reset_shifts=True)
    assert np.array_equal(ms.cpu().numpy(), g[f"{tag}_mask_post"])
    sub = xs[:, 1, :, :: max(1, cfg.img_size[0] // 16), :: max(1, cfg.img_size[0] // 16)].cpu().numpy()
    assert np.array_equal(sub, g[f"{tag}_x_frame1_sub"])
    d = xs.double()
    assert np.allclose([d.sum().item(), (d * d).sum().item()], g[f"{tag}_x_digest"], rtol=1e-12)
    if tag == "tiny":
        assert np.array_equal(xs.cpu().numpy(), g["tiny_x_shift"])
    assert len(G.shifts) == len(shifts) and [tuple(int(v) for v in s) for s in G.shifts] == shifts
    # frame 0 is untouched and every prompt's frame 1 differs from the static frame only at destination patches
    assert torch.equal(xs[:, 0], x[:, 0].expand(xs.shape[0], -1, -1, -1))


def test_shift_prompts_match_oracle_random_tables():
    cfg = TINY
    G = wrapper(cfg)
    g = torch.Generator().manual_seed(5)
    B, S_, n = 2, 7, 16
    x = torch.rand(B, 2, 3, 32, 32, generator=g)
    active = torch.ones(B, 2 * n, S_, dtype=torch.bool)
    active[:, :n] = False
    passive = active.clone()
    for b in range(B):
        for s in range(S_):
            active[b, n + int(torch.randint(n, (1,), generator=g)), s] = False
            passive[b, n + int(torch.randint(n, (1,), generator=g)), s] = False
    shifts = [(int(torch.randint(-4, 5, (1,), generator=g)), int(torch.randint(-4, 5, (1,), generator=g))) for _ in range(B * S_)]
    G.mask_rectangularizer.set_mode(None)
    xs, ms = G.create_motion_counterfactuals(x.cuda(), masks=passive.cuda(), active_patches=active.cuda(), shifts=shifts, reset_shifts=True)
    xo, mo = O.create_motion_counterfactuals(x, passive, active, shifts, 8)
    assert torch.equal(xs.cpu(), xo) and torch.equal(ms.cpu(), mo)


def test_counterfactual_prediction_driver_chunking_invariance():
    cfg = TINY
    G = wrapper(cfg)
    x = torch.from_numpy(S.synthetic_frames(1, cfg, 21))[:, 0].cuda()  # single image -> static 2-frame movie
    n = 16
    S_ = 6
    active = torch.ones(1, 2 * n, S_, dtype=torch.bool)
    active[:, :n] = False
    for s in range(S_):
        active[0, n + (5 + s) % n, s] = False
    shifts = [(1, 0), (0, 1), (-1, 0), (0, -1), (1, 1), (-1, 2)]
    # prompt 5 moves its patch out of the frame, so RectangularizeMasks un-masks a random patch (global torch RNG)
    torch.manual_seed(11)
    ya = G.predict_counterfactual_videos(x, active.cuda(), shifts=shifts, sample_batch_size=6)
    torch.manual_seed(11)
    yb = G.predict_counterfactual_videos(x, active.cuda(), shifts=shifts, sample_batch_size=2)
    assert ya.shape == (S_, 2, 3, 32, 32)
    assert (ya - yb).abs().max().item() <= 1e-5  # result independent of sample_batch_size (f-2)
    # against the oracle end to end
    W = {k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 3).items()}
    xo, mo = O.create_motion_counterfactuals(x.cpu()[:, None].expand(-1, 2, -1, -1, -1), torch.ones(1, 2 * n, S_, dtype=torch.bool) & ~(torch.arange(2 * n) < n)[None, :, None],
                                             active, shifts, 8)
    torch.manual_seed(11)
    mo = O.rectangularize_masks_min(mo)
    with torch.no_grad():
        ref = O.predict(W, TINY_SPEC, xo, mo, frame=None)
    assert (ya.cpu() - ref).abs().max().item() <= 2e-4
