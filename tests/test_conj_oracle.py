"""Oracle of the IMU-conditioned conjoined padded predictor (oracle/conj_oracle.py) against the reference
outputs captured in tests/golden/conj_*.npz, and the host mirror's schema."""
import os

import numpy as np
import torch

from counterfactualworldmodels_amd import config as C, synthetic as S
from oracle import conj_oracle as CO, vmae_oracle as V

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TINY_MAIN = C.VmaeConfig(name="tiny_conj_main", img_size=(32, 32), patch=4, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128, dec_depth=1, dec_heads=2)
TINY_CONJ = C.ConjConfig(name="tiny_conj", main=TINY_MAIN, main_max_pad=8, ctx_seq_len=64, ctx_enc_dim=64, ctx_dec_dim=64, ctx_enc_heads=2,
                         ctx_dec_heads=2, ctx_max_pad=4, enc_cross=(0,), dec_cross=(0,))
TINY_SPEC = CO.ConjSpec(main=V.VmaeSpec(img_size=(32, 32), patch=4, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128, dec_depth=1, dec_heads=2),
                        main_max_pad=8, ctx_seq_len=64, ctx_enc_dim=64, ctx_dec_dim=64, ctx_enc_heads=2, ctx_dec_heads=2, ctx_max_pad=4,
                        enc_cross=(0,), dec_cross=(0,))


def conj_weights(cfg, seed, sharp=False):
    w = {k: S.synthetic_tensor(k, shp, seed) for k, shp in C.conj_state_dict_schema(cfg).items()}
    if sharp:
        w = S.sharpen_state_dict(w, seed)
    return {k: torch.from_numpy(v) for k, v in w.items()}


def test_schema_known_answers():
    cfg = C.CONJ_CONFIGS["imu400_base_4x4patch_2frames_1tube"]
    sch = C.conj_state_dict_schema(cfg)
    assert len(sch) == 634
    assert sum(int(np.prod(s)) for s in sch.values()) == 148_265_040  # ipynb:375
    assert sch["context_stream.encoder.patch_embed.proj.weight"] == (384, 6, 16, 1, 1)
    assert sch["encoder_conjoining_blocks.9-9.cross_attention.qk_src.weight"] == (1536, 384)
    assert sch["decoder_conjoining_blocks.3-3.mlp.src.layers.2.weight"] == (192, 384)


def test_padding_rule_known_answer():
    """SURVEY.md A7 probe: Nt=8, P=4, visible counts 5/7/4."""
    mask = torch.ones(3, 8, dtype=torch.bool)
    mask[0, :5] = False
    mask[1, :7] = False
    mask[2, :4] = False
    full, null = CO.padding_masks(mask, 4)
    assert full[:, 8:].int().tolist() == [[0, 0, 1, 1], [1, 1, 1, 1], [0, 0, 0, 1]]
    assert null.int().tolist() == [[0, 0, 0, 1, 1], [0, 1, 1, 1, 1], [0, 0, 0, 0, 1]]


def test_tiny_conj_vs_reference():
    g = np.load(os.path.join(GOLDEN, "conj_tiny.npz"))
    W = conj_weights(TINY_CONJ, int(g["seed"]))
    x, mask, imu, mc = (torch.from_numpy(g[k]) for k in ("x", "mask", "imu", "mask_context"))
    with torch.no_grad():
        y = CO.conj_forward(W, TINY_SPEC, V.preprocess(x), mask, imu, mc)
        mask_eq = torch.from_numpy(g["mask_eq"])
        mc_eq = torch.zeros(2, TINY_CONJ.ctx_tokens, dtype=torch.bool)
        y_eq = CO.conj_forward(W, TINY_SPEC, V.preprocess(x[:2]), mask_eq, imu[:2], mc_eq)
        video = CO.predict(W, TINY_SPEC, x[:2], mask_eq, imu[:2], mc_eq, frame=None)
    assert y.shape == g["y_tokens"].shape and np.abs(y.numpy() - g["y_tokens"]).max() <= 1e-5
    assert np.abs(y_eq.numpy() - g["y_tokens_eq"]).max() <= 1e-5
    assert (y_eq[:, -TINY_CONJ.main_max_pad:] == 0).all()  # equal counts: the last P rows are the (zeroed) pad slots
    assert np.abs(video.numpy() - g["video_eq"]).max() <= 1e-5
    # ragged rows: exactly P - (vmax - v_b) zero rows per sample
    zero_rows = (y.abs().sum(-1) == 0).sum(-1).tolist()
    vis = (~mask).sum(-1)
    assert zero_rows == [int(8 - (vis.max() - v)) for v in vis]


def test_imu400_full_size_vs_reference():
    g = np.load(os.path.join(GOLDEN, "conj_imu400_b2.npz"))
    cfg = C.CONJ_CONFIGS["imu400_base_4x4patch_2frames_1tube"]
    W = conj_weights(cfg, int(g["seed"]))
    x = torch.from_numpy(S.synthetic_frames(2, cfg.main, 0))
    mask, imu = torch.from_numpy(g["mask"]), torch.from_numpy(g["imu"])
    with torch.no_grad():
        y = CO.conj_forward(W, CO.IMU400_BASE_4X4, V.preprocess(x), mask, imu, torch.zeros(2, 25, dtype=torch.bool))
    assert y.shape == (2, 6272 + 64 - 3142, 48)
    err = np.abs(y.numpy() - g["y_tokens"]).max()
    assert err <= 5e-5, err


def test_imu400_sharp_weights_vs_reference():
    """Hostile weights (synthetic.sharpen_state_dict) through the full-size IMU-conditioned model, one masked context token."""
    g = np.load(os.path.join(GOLDEN, "conj_imu400_sharp_b1.npz"))
    cfg = C.CONJ_CONFIGS["imu400_base_4x4patch_2frames_1tube"]
    seed = int(g["seed"])
    W = conj_weights(cfg, seed, sharp=True)
    x = torch.from_numpy(S.synthetic_frames(1, cfg.main, seed))
    mask, imu, mc = (torch.from_numpy(g[k]) for k in ("mask", "imu", "mask_context"))
    assert np.array_equal(mask.numpy(), S.synthetic_masks(1, cfg.main, 4, seed))
    with torch.no_grad():
        y = CO.conj_forward(W, CO.IMU400_BASE_4X4, V.preprocess(x), mask, imu, mc)
    err = np.abs(y.numpy() - g["y_tokens"]).max()
    assert y.shape == g["y_tokens"].shape and err <= 1e-4, err


def test_context_output_vs_reference():
    """`forward(..., output_context=True)` (conjoined_vmae.py:852-887, 990-1011): the context stream's predictions, tiny model with ragged
    visible counts and masked context tokens, and the full-size model with three masked IMU tokens."""
    g = np.load(os.path.join(GOLDEN, "conj_tiny_ctx.npz"))
    W = conj_weights(TINY_CONJ, int(g["seed"]))
    x, mask, imu, mc = (torch.from_numpy(g[k]) for k in ("x", "mask", "imu", "mask_context"))
    with torch.no_grad():
        y, y_c = CO.conj_forward(W, TINY_SPEC, V.preprocess(x), mask, imu, mc, output_context=True)
    assert y.shape == g["y_tokens"].shape and np.abs(y.numpy() - g["y_tokens"]).max() <= 1e-5
    assert y_c.shape == g["y_ctx_tokens"].shape == (3, TINY_CONJ.ctx_tokens + TINY_CONJ.ctx_max_pad - 4, TINY_CONJ.ctx_out_dim)
    assert np.abs(y_c.numpy() - g["y_ctx_tokens"]).max() <= 1e-5
    assert np.array_equal((y_c.abs().sum(-1) == 0).numpy(), g["ctx_null_mask"])   # zero rows = the masked pad slots, nothing else
    g = np.load(os.path.join(GOLDEN, "conj_imu400_ctx_b1.npz"))
    cfg = C.CONJ_CONFIGS["imu400_base_4x4patch_2frames_1tube"]
    W = conj_weights(cfg, int(g["seed"]))
    x = torch.from_numpy(S.synthetic_frames(1, cfg.main, int(g["frames_seed"])))
    mask, imu, mc = (torch.from_numpy(g[k]) for k in ("mask", "imu", "mask_context"))
    with torch.no_grad():
        y, y_c = CO.conj_forward(W, CO.IMU400_BASE_4X4, V.preprocess(x), mask, imu, mc, output_context=True)
    assert y_c.shape == g["y_ctx_tokens"].shape == (1, 28, 96)
    assert np.abs(y_c.numpy() - g["y_ctx_tokens"]).max() <= 5e-5
    assert np.abs(y[:, :8].numpy() - g["y_tokens_head"]).max() <= 5e-5
    assert (y_c[0, 3:] == 0).all() and (y_c[0, :3].abs().sum(-1) > 0).all()   # 3 masked IMU tokens, then the 25 (masked) pad slots


def test_host_mirror_schema_and_attributes():
    from counterfactualworldmodels_amd import conjoined_vmae as CV

    m = CV.ConjoinedPaddedVisionTransformer(TINY_CONJ)
    sch = C.conj_state_dict_schema(TINY_CONJ)
    sd = m.state_dict()
    assert list(sd) == list(sch) and all(tuple(sd[k].shape) == sch[k] for k in sd)
    assert not hasattr(m, "padding_mask")  # like the reference before a forward (conjoined_vmae.py:347-354)
    assert m.main_stream.max_padding_tokens == 8 and m.context_stream.max_padding_tokens == 4
    assert m.patch_size == (1, 4, 4) and m.mask_size == (2, 8, 8) and m.context_stream.encoder.num_tokens == 4
    assert callable(CV.imu400_base_4x4patch_2frames_1tube)
