"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (this container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--skip-large]

The reference's own tests pin nothing for the predictor path (there are none), so
the oracle (`oracle/vmae_oracle.py`) and the HIP path are pinned against these
captured reference outputs.  Weights/inputs come from the package's deterministic
generator (`counterfactualworldmodels_amd/synthetic.py`), loaded into the *reference*
modules; only inputs that are cheap to store, and the reference outputs, are saved.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_import  # noqa: E402
from counterfactualworldmodels_amd import config as C  # noqa: E402
from counterfactualworldmodels_amd import synthetic as S  # noqa: E402

S_ = S

TINY = C.VmaeConfig(
    name="tiny_8x8", img_size=(32, 32), patch=8, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128, dec_depth=1, dec_heads=2
)


def build_ref_model(ns, cfg: C.VmaeConfig, seed: int, sharp: bool = False):
    from functools import partial

    if cfg.name in ("base_8x8patch_2frames_1tube", "large_4x4patch_2frames_1tube", "base_16x16patch_2frames_1tube"):
        m = getattr(ns.vmae, cfg.name)()
    else:
        m = ns.vmae.PretrainVisionTransformer(
            img_size=cfg.img_size[0],
            patch_size=(cfg.patch, cfg.patch),
            encoder_embed_dim=cfg.enc_dim,
            encoder_depth=cfg.enc_depth,
            encoder_num_heads=cfg.enc_heads,
            encoder_num_classes=0,
            decoder_embed_dim=cfg.dec_dim,
            decoder_depth=cfg.dec_depth,
            decoder_num_heads=cfg.dec_heads,
            mlp_ratio=cfg.mlp_ratio,
            qkv_bias=True,
            num_frames=cfg.num_frames,
            tubelet_size=1,
            norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
        )
    sd = {k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, seed, sharp=sharp).items()}
    res = m.load_state_dict(sd)
    assert not res.missing_keys and not res.unexpected_keys, res
    return m.eval().requires_grad_(False)


def run_model_case(ns, cfg, *, batch, k_vis, clump, seed, out_name, through_wrapper=True, sharp=False):
    t0 = time.time()
    m = build_ref_model(ns, cfg, seed, sharp)
    x = torch.from_numpy(S.synthetic_frames(batch, cfg, seed))
    mask = torch.from_numpy(S.synthetic_masks(batch, cfg, k_vis, seed, clump))
    out = {
        "cfg_name": np.array(cfg.name),
        "seed": np.array(seed),
        "batch": np.array(batch),
        "k_vis": np.array(k_vis),
        "clump": np.array(clump),
        "mask": mask.numpy(),
        "sharp": np.array(sharp),
    }
    with torch.no_grad():
        if through_wrapper:
            # the reference "Predictor class surface": prediction.py:17, predict :406
            G = ns.prediction.PredictorBasedGenerator(
                predictor=m, imagenet_normalize_inputs=True, temporal_dim=2, seed=0
            )
            y_model = m(G._preprocess(x), mask.clone())
            video = G.predict(x, mask.clone(), frame=None)
            out["video_digest"] = np.array(
                [video.double().sum().item(), video.double().abs().sum().item(), (video.double() ** 2).sum().item()]
            )
            out["video_frame1_rows"] = video[:, 1, :, :: max(1, cfg.img_size[0] // 8)].numpy().copy()
        else:
            mean = torch.tensor(C.IMAGENET_MEAN).view(1, 3, 1, 1, 1)
            std = torch.tensor(C.IMAGENET_STD).view(1, 3, 1, 1, 1)
            y_model = m((x.transpose(1, 2) - mean) / std, mask.clone())
    out["y_tokens"] = y_model.numpy()
    if cfg.name.startswith("tiny_"):
        out["x"] = x.numpy()
    np.savez_compressed(os.path.join(HERE, out_name), **out)
    print(f"[golden] {out_name}: y {tuple(y_model.shape)} std {y_model.std():.4f} ({time.time() - t0:.1f}s)")


def run_block_case(ns):
    """Per-op tiles at real width, short N."""
    from functools import partial

    D, H, N, B = 768, 12, 40, 2
    blk = ns.vutils.Block(dim=D, num_heads=H, mlp_ratio=4, qkv_bias=True, init_values=0.0,
                          norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
    schema = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
    sd = {k: torch.from_numpy(S.synthetic_tensor("golden_block." + k, shp, 7)) for k, shp in schema.items()}
    blk.load_state_dict(sd)
    blk.eval()
    g = np.random.Generator(np.random.PCG64(11))
    x = torch.from_numpy(g.standard_normal((B, N, D), dtype=np.float32))
    with torch.no_grad():
        h1 = blk.norm1(x)
        a = blk.attn(h1)
        y = blk(x)
        mlp = blk.mlp(blk.norm2(x + a))
    np.savez_compressed(
        os.path.join(HERE, "block_768.npz"),
        x=x.numpy(), norm1=h1.numpy(), attn=a.numpy(), mlp=mlp.numpy(), y=y.numpy(),
        heads=np.array(H), seed=np.array(7),
    )
    print("[golden] block_768.npz")


def run_index_cases(ns):
    """Bit-exact mask / index fixtures (part)."""
    out = {}
    # RectangularizeMasks('min') with torch's global RNG (masking.py:90-132)
    rect = ns.masking.RectangularizeMasks("min")
    g = np.random.Generator(np.random.PCG64(5))
    masks = g.random((6, 2 * 14 * 14)) < 0.9
    masks[:, :196] = False
    torch.manual_seed(1234)
    m_in = torch.from_numpy(masks.copy())
    m_out = rect(m_in.clone())
    out["rect_in"] = masks
    out["rect_out"] = m_out.numpy()
    out["rect_seed"] = np.array(1234)
    # sinusoid tables (VideoMAE/utils.py:251-268) and torch-fp32 pos_embedding (transformer.py:37-52)
    for (n, d) in [(1568, 768), (1568, 384), (6272, 1024), (6272, 512), (32, 128)]:
        t = ns.vutils.get_sinusoid_encoding_table(n, d)[0].double()
        out[f"sinusoid_{n}_{d}"] = np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item(),
                                             t[n - 1, d - 1].item(), t[n // 2, d // 3].item()])
    for (n, d) in [(25, 384), (25, 192), (6272, 512)]:
        t = ns.transformer.pos_embedding(n, d, "cpu")[0].double()
        out[f"posemb_{n}_{d}"] = np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item(),
                                           t[n - 1, d - 1].item(), t[n // 2, d // 3].item()])
    # Patchify round trip / pred_patches_to_video (prediction.py:245-259)
    cfg = TINY
    m = build_ref_model(ns, cfg, 3)
    G = ns.prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2, seed=0)
    x = torch.from_numpy(S.synthetic_frames(3, cfg, 3))
    mask = torch.from_numpy(S.synthetic_masks(3, cfg, 4, 3))
    G.set_image_size(x.shape[-2:])
    G.inp_shape = x.shape
    y = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).standard_normal(
        (3, int(mask[0].sum()), cfg.out_dim), dtype=np.float32))
    vid = G.pred_patches_to_video(y, x, mask)
    out["unembed_x"] = x.numpy()
    out["unembed_mask"] = mask.numpy()
    out["unembed_y"] = y.numpy()
    out["unembed_video"] = vid.numpy()
    np.savez_compressed(os.path.join(HERE, "index_ops.npz"), **out)
    print("[golden] index_ops.npz")


def run_shift_cases(ns):
    """f-1 fixtures: the reference's own `FlowGenerator.create_motion_counterfactuals` (segmentation.py:278-344)
    on a 4x4 grid (tiny) and on the 28x28 B/8 grid, incl. negative and out-of-frame shifts and passive patches."""
    assert ns.segmentation is not None, getattr(ns, "segmentation_error", None)

    class DummyFlow(torch.nn.Module):
        def forward(self, x, *a, **k):
            return torch.zeros(x.shape[0], x.shape[1] - 1, 2, *x.shape[-2:])

    out = {}
    for tag, cfg, S in (("tiny", TINY, 12), ("base8", C.CONFIGS["base_8x8patch_2frames_1tube"], 24)):
        m = build_ref_model(ns, cfg, 3) if tag == "tiny" else ns.vmae.base_8x8patch_2frames_1tube().eval()
        G = ns.segmentation.FlowGenerator(predictor=m, flow_model=DummyFlow(), imagenet_normalize_inputs=True, temporal_dim=2, seed=0)
        B = 1  # the reference loop indexes shifts[i] for i < B*S with len(shifts) == S: it only supports B == 1
        gh = cfg.img_size[0] // cfg.patch
        n = gh * gh
        x = torch.from_numpy(S_.synthetic_frames(B, cfg, 21))
        g = np.random.Generator(np.random.PCG64(31))
        active = torch.ones(B, 2 * n, S, dtype=torch.bool)
        active[:, :n] = False
        passive = torch.ones(B, 2 * n, S, dtype=torch.bool)
        passive[:, :n] = False
        shifts = []
        for b in range(B):
            for s in range(S):
                a = int(g.integers(n))
                active[b, n + a, s] = False
                if s % 3 == 0:  # one passive patch as well
                    pz = int(g.integers(n))
                    passive[b, n + pz, s] = False
                if s % 5 == 4:  # two active patches
                    active[b, n + int(g.integers(n)), s] = False
        lim = gh if tag == "tiny" else 3
        for s in range(S):
            while True:
                dy, dx = int(g.integers(-lim, lim + 1)), int(g.integers(-lim, lim + 1))
                if dy or dx:
                    break
            shifts.append([dy, dx])
        G.set_input(x[:, 0:2])
        G.shifter.set_shapes(x, mask=active[..., 0])
        torch.manual_seed(77)
        x_shift, mask_post = G.create_motion_counterfactuals(
            x, masks=passive, active_patches=active, shifts=[list(v) for v in shifts], num_samples=S, fix_passive=True, reset_shifts=True)
        out[f"{tag}_active"] = active.numpy()
        out[f"{tag}_passive"] = passive.numpy()
        out[f"{tag}_shifts"] = np.array(shifts, dtype=np.int32)
        out[f"{tag}_mask_post"] = mask_post.numpy()
        out[f"{tag}_rect_seed"] = np.array(77)
        xs = x_shift.double()
        out[f"{tag}_x_digest"] = np.array([xs.sum().item(), (xs * xs).sum().item()])
        out[f"{tag}_x_frame1_sub"] = x_shift[:, 1, :, :: max(1, cfg.img_size[0] // 16), :: max(1, cfg.img_size[0] // 16)].numpy().copy()
        if tag == "tiny":
            out["tiny_x_shift"] = x_shift.numpy()
    np.savez_compressed(os.path.join(HERE, "shift_prompts.npz"), **out)
    print("[golden] shift_prompts.npz")


TINY_CONJ_MAIN = C.VmaeConfig(name="tiny_conj_main", img_size=(32, 32), patch=4, enc_dim=128, enc_depth=2, enc_heads=2,
                               dec_dim=128, dec_depth=1, dec_heads=2)
TINY_CONJ = C.ConjConfig(name="tiny_conj", main=TINY_CONJ_MAIN, main_max_pad=8, ctx_seq_len=64, ctx_enc_dim=64, ctx_dec_dim=64,
                         ctx_enc_heads=2, ctx_dec_heads=2, ctx_max_pad=4, enc_cross=(0,), dec_cross=(0,))


def build_ref_conj(ns, cfg: C.ConjConfig, seed: int, sharp: bool = False):
    from functools import partial

    conj = ns.conj
    if cfg.name == "imu400_base_4x4patch_2frames_1tube":
        m = conj.imu400_base_4x4patch_2frames_1tube()
    else:
        mc = cfg.main
        main_kw = dict(encoder_func=ns.vmae.PretrainVisionTransformerEncoder, tubelet_size=1, decoder_num_classes=None,
                       min_padding_tokens=0, max_padding_tokens=cfg.main_max_pad)
        ctx_kw = dict(encoder_func=conj.ImuEncoder, tubelet_size=cfg.ctx_tubelet, spacetime_separable_pos_embed=True,
                      encoder_embed_dim=cfg.ctx_enc_dim, decoder_embed_dim=cfg.ctx_dec_dim, sequence_length=cfg.ctx_seq_len,
                      decoder_num_classes=cfg.ctx_out_dim, min_padding_tokens=0, max_padding_tokens=cfg.ctx_max_pad,
                      concat_dummy_token=False)
        m = conj.ConjoinedPaddedVisionTransformer(
            img_size=mc.img_size[0], patch_size=(mc.patch, mc.patch), encoder_embed_dim=mc.enc_dim, encoder_depth=mc.enc_depth,
            encoder_num_heads=mc.enc_heads, encoder_num_classes=0, decoder_embed_dim=mc.dec_dim, decoder_num_heads=mc.dec_heads,
            decoder_depth=mc.dec_depth, mlp_ratio=4, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
            main_model_func=conj.PaddedVisionTransformer, main_model_kwargs=main_kw, main_input="rgb01",
            main_input_kwargs={"unnormalize": False}, context_model_func=conj.PaddedVisionTransformer, context_model_kwargs=ctx_kw,
            context_input="imu", conjoin_encoder_layers=list(cfg.enc_cross), conjoin_decoder_layers=True)
    sch = C.conj_state_dict_schema(cfg)
    sd = m.state_dict()
    assert list(sch) == list(sd) and all(tuple(sd[k].shape) == sch[k] for k in sd)
    weights = {k: S.synthetic_tensor(k, shp, seed) for k, shp in sch.items()}
    if sharp:
        weights = S.sharpen_state_dict(weights, seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    return m.eval().requires_grad_(False)


def run_conj_cases(ns, skip_large=False):
    """BASELINE configs[4]: the IMU-conditioned conjoined padded predictor (conjoined_vmae.py:889-1011, 1230-1243)."""
    # ---- tiny, ragged visible counts in both streams, through the model and through the reference wrapper
    cfg = TINY_CONJ
    m = build_ref_conj(ns, cfg, 5)
    g = np.random.Generator(np.random.PCG64(2))
    B, n = 3, cfg.main.tokens_per_frame
    x = torch.from_numpy(g.random((B, 2, 3, 32, 32), dtype=np.float32))
    mask = torch.zeros(B, 2 * n, dtype=torch.bool)
    mask[:, n:] = True
    for b, vis in enumerate(([3, 9], [5], [6, 7, 16])):
        mask[b, [n + v for v in vis]] = False
    imu = torch.from_numpy((g.standard_normal((B, 6, cfg.ctx_seq_len)) * 0.1).astype(np.float32))
    mc = torch.zeros(B, cfg.ctx_tokens, dtype=torch.bool)
    mc[0, 1] = True
    mc[1, 0] = True
    mc[1, 3] = True
    G = ns.prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2, seed=0)
    out = {"x": x.numpy(), "mask": mask.numpy(), "imu": imu.numpy(), "mask_context": mc.numpy(), "seed": np.array(5)}
    with torch.no_grad():
        m._reset_padding_mask()
        out["y_tokens"] = m(G._preprocess(x), mask.clone(), x_context=imu, mask_context=mc).numpy()
        # the padding state the reference leaves behind until the wrapper resets it (conjoined_vmae.py:49-116): SURVEY.md 8c-4
        pad = {"padding_mask": m.main_stream.padding_mask.numpy(), "full_input_mask": m.main_stream.full_input_mask.numpy(),
               "null_mask": m.main_stream.null_mask.numpy(), "ctx_padding_mask": m.context_stream.padding_mask.numpy()}
        m._reset_padding_mask()
        # equal visible counts (the normal case after RectangularizeMasks): no visible pads, last P rows are zero
        mask_eq = torch.from_numpy(S.synthetic_masks(2, cfg.main, 3, 9))
        mc_eq = torch.zeros(2, cfg.ctx_tokens, dtype=torch.bool)
        out["mask_eq"] = mask_eq.numpy()
        out["y_tokens_eq"] = m(G._preprocess(x[:2]), mask_eq.clone(), x_context=imu[:2], mask_context=mc_eq).numpy()
        pad.update({"padding_mask_eq": m.main_stream.padding_mask.numpy(), "full_input_mask_eq": m.main_stream.full_input_mask.numpy(),
                    "null_mask_eq": m.main_stream.null_mask.numpy()})
        np.savez_compressed(os.path.join(HERE, "conj_padding.npz"), **pad)
        m._reset_padding_mask()
        torch.manual_seed(3)
        video = G.predict(x[:2], mask_eq.clone(), frame=None, x_context=imu[:2], mask_context=mc_eq)
        out["video_eq"] = video.numpy()
    np.savez_compressed(os.path.join(HERE, "conj_tiny.npz"), **out)
    print("[golden] conj_tiny.npz", out["y_tokens"].shape, out["video_eq"].shape)
    if skip_large:
        return
    # ---- full size, B=2 with 4 and 6 visible frame-2 patches (exercises the null-token padding)
    cfg = C.CONJ_CONFIGS["imu400_base_4x4patch_2frames_1tube"]
    t0 = time.time()
    m = build_ref_conj(ns, cfg, 0)
    x = torch.from_numpy(S.synthetic_frames(2, cfg.main, 0))
    mask = torch.from_numpy(np.stack([S.synthetic_masks(1, cfg.main, 4, 0)[0], S.synthetic_masks(1, cfg.main, 6, 1)[0]]))
    imu = torch.from_numpy((np.random.Generator(np.random.PCG64(7)).standard_normal((2, 6, 400)) * 0.1).astype(np.float32))
    mc = torch.zeros(2, 25, dtype=torch.bool)
    mean = torch.tensor(C.IMAGENET_MEAN).view(1, 3, 1, 1, 1)
    std = torch.tensor(C.IMAGENET_STD).view(1, 3, 1, 1, 1)
    with torch.no_grad():
        y = m((x.transpose(1, 2) - mean) / std, mask.clone(), x_context=imu, mask_context=mc)
    np.savez_compressed(os.path.join(HERE, "conj_imu400_b2.npz"), mask=mask.numpy(), imu=imu.numpy(), y_tokens=y.numpy(), seed=np.array(0))
    print(f"[golden] conj_imu400_b2.npz {tuple(y.shape)} std {y.std():.4f} ({time.time() - t0:.1f}s)")


def run_wrapper_cases(ns):
    """Wrapper-surface fixtures (SURVEY.md §8b "wrapper surface consumers rely on", §8c-4): mask generators, patch-index masks,
    masked-patch compositing, the single-prompt counterfactual of the UI's click handler, `predict_error`, and the batch driver
    `predict_counterfactual_videos_and_flows` incl. the IMU keyword forwarding -- every output produced by the reference classes."""
    out = {}
    # ---- mask generators (masking.py:267-401, 478-545); known answer: 3104 masked on (2,56,56) (demo notebook)
    x2 = torch.zeros(2, 2, 3, 8, 8)
    gen = ns.masking.RotatedTableUniformMaskingGenerator(input_size=(2, 56, 56), mask_ratio=0.99, clumping_factor=2,
                                                         randomize_num_visible=False, always_batch=True, seed=0)
    out["gen_rot56_none"] = gen(None).numpy()
    out["gen_rot56_b2"] = gen(x2).numpy()
    assert int(out["gen_rot56_none"].sum()) == 3104
    gen = ns.masking.RotatedTableUniformMaskingGenerator(input_size=(3, 7, 7), mask_ratio=0.75, clumping_factor=2, seed=1,
                                                         randomize_num_visible=True)
    out["gen_rot7_pad_b2"] = np.stack([gen(x2).numpy() for _ in range(3)])
    gen = ns.masking.RotatedTableUniformMaskingGenerator(input_size=(2, 8, 8), mask_ratio=0.5, seed=4, full_mask_prob=0.5)
    out["gen_rot8_full_b5"] = np.stack([gen(torch.zeros(5, 1)).numpy() for _ in range(2)])
    gen = ns.masking.MaskingGenerator(input_size=(1, 7, 7), mask_ratio=0.75, clumping_factor=2, seed=1, visible_frames=1)
    out["gen_base7_b2"] = gen(x2).numpy()
    gen = ns.masking.RotatedTableUniformMaskingGenerator(input_size=(2, 28, 28), mask_ratio=0.9, seed=2)
    gen.num_visible = 3
    out["gen_rot28_nv3"] = gen(x2).numpy()

    # ---- tiny predictor behind the reference FlowGenerator
    class DummyFlow(torch.nn.Module):
        def forward(self, x, backward=False, **k):
            d = x[:, 1:] - x[:, :-1]
            return torch.stack([d.mean(2), d.amax(2)], 2)  # [R,T-1,2,H,W]

    cfg = TINY
    m = build_ref_model(ns, cfg, 3)
    G = ns.segmentation.FlowGenerator(predictor=m, flow_model=DummyFlow(), imagenet_normalize_inputs=True, temporal_dim=2, seed=0)
    x = torch.from_numpy(S.synthetic_frames(2, cfg, 41))
    n = cfg.tokens_per_frame
    G.set_input(x)
    # generate_mask_from_patch_idx_list (prediction.py:640-649): image coordinates, (h,w) / (t,h,w) / (b,t,h,w) entries
    out["idx_hw"] = G.generate_mask_from_patch_idx_list([[9, 17], [31, 2]], b=1, frame=-1).numpy()
    out["idx_thw"] = G.generate_mask_from_patch_idx_list([[1, 9, 17], [1, 24, 24]], b=0, frame=1).numpy()
    out["idx_bthw"] = G.generate_mask_from_patch_idx_list([[0, 1, 9, 17], [1, 1, 31, 2]], frame=1, stride=8).numpy()
    # get_masked_pred_patches (prediction.py:261-283)
    g = np.random.Generator(np.random.PCG64(3))
    preds = torch.from_numpy(g.random((2, 1, 3, 32, 32), dtype=np.float32))
    pmask = torch.from_numpy(g.random((2, n)) < 0.5)
    out["mpp_preds"], out["mpp_mask"] = preds.numpy(), pmask.numpy()
    out["mpp_plain"] = G.get_masked_pred_patches(preds, pmask).numpy()
    out["mpp_invert_fill"] = G.get_masked_pred_patches(preds, pmask, invert=True, fill_value=[0.1, 0.2, 0.3]).numpy()
    out["mpp_fill_tensor"] = G.get_masked_pred_patches(preds, pmask, fill_value=1 - preds).numpy()
    # predict_error (prediction.py:331-343)
    mask = torch.from_numpy(S.synthetic_masks(2, cfg, 4, 41))
    with torch.no_grad():
        out["err_frame1"] = G.predict_error(x, mask.clone(), frame=1).numpy()
        out["err_all"] = G.predict_error(x, mask.clone(), frame=None).numpy()
    out["err_mask"] = mask.numpy()
    # get_counterfactual_prediction (prediction.py:781-812): the UI's single prompt (interface.py:273-299)
    img = x[:1, 0]
    passive = torch.ones(1, 2 * n, dtype=torch.bool)
    passive[:, :n] = False
    passive[0, n + 3] = False
    active = torch.ones(1, 2 * n, dtype=torch.bool)
    active[:, :n] = False
    active[0, n + 6] = False
    with torch.no_grad():
        xs = G.make_static_movie(img[:, None], T=2)
        G.shifts = None
        x_p, mask_p = G._shift(xs, passive.clone(), active_patches=active.clone(), shift=(1, -1), frame=1)
        G.shifts = None
        y_p = G.get_counterfactual_prediction(img, mask=passive.clone(), active_patches=active.clone(), shift=(1, -1))
    out["cf_img"], out["cf_passive"], out["cf_active"] = img.numpy(), passive.numpy(), active.numpy()
    out["cf_x_p"], out["cf_mask_p"], out["cf_y"] = x_p.numpy(), mask_p.numpy(), y_p.numpy()
    out["cf_shifts"] = np.array(G.shifts)
    # fix_passive=True on a movie whose frames differ: MakeStatic on the passive patches before the shift (prediction.py:802-803,
    # perturbation.py:120-145); two passive patches, one of them the shift's destination's neighbour
    movie = x[:1].clone()
    passive2 = passive.clone()
    passive2[0, n + 10] = False
    with torch.no_grad():
        ms_x, ms_mask = G.make_static(movie, passive2.clone())
        G.shifts = None
        x_p2, mask_p2 = G._shift(ms_x, passive2.clone(), active_patches=active.clone(), shift=(1, -1), frame=1)
        G.shifts = None
        y_p2 = G.get_counterfactual_prediction(movie, mask=passive2.clone(), active_patches=active.clone(), shift=(1, -1), fix_passive=True)
    out["ms_movie"], out["ms_passive"], out["ms_x"], out["ms_mask"] = movie.numpy(), passive2.numpy(), ms_x.numpy(), ms_mask.numpy()
    out["ms_x_p"], out["ms_mask_p"], out["ms_y"] = x_p2.numpy(), mask_p2.numpy(), y_p2.numpy()
    # the batch driver with a flow model (segmentation.py:346-432): ordering, shifts list, flow shape
    S_n = 5
    act = torch.ones(1, 2 * n, S_n, dtype=torch.bool)
    act[:, :n] = False
    for s in range(S_n):
        act[0, n + (2 + 3 * s) % n, s] = False
    shifts = [[1, 0], [0, 1], [-1, 0], [0, -1], [1, 1]]
    torch.manual_seed(5)
    with torch.no_grad():
        ys, fs = G.predict_counterfactual_videos_and_flows(img, active_patches=act.clone(), shifts=[list(v) for v in shifts],
                                                           num_samples=S_n, sample_batch_size=64)
    out["drv_active"], out["drv_shifts"] = act.numpy(), np.array(shifts, dtype=np.int32)
    out["drv_ys"], out["drv_flows"], out["drv_shift_list"] = ys.numpy(), fs.numpy(), np.array(G.shifts)
    np.savez_compressed(os.path.join(HERE, "wrapper_surface.npz"), **out)
    print("[golden] wrapper_surface.npz", {k: v.shape for k, v in out.items() if k.startswith(("cf_", "drv_"))})

    # ---- IMU-conditioned driver: the reference base class forwards x_context / mask_context per chunk (B = 1)
    cfgc = TINY_CONJ
    mc_ = build_ref_conj(ns, cfgc, 5)
    Gc = ns.segmentation.FlowGenerator(predictor=mc_, flow_model=DummyFlow(), imagenet_normalize_inputs=True, temporal_dim=2, seed=0)
    nc = cfgc.main.tokens_per_frame
    g = np.random.Generator(np.random.PCG64(8))
    imgc = torch.from_numpy(g.random((1, 3, 32, 32), dtype=np.float32))
    imu = torch.from_numpy((g.standard_normal((1, 6, cfgc.ctx_seq_len)) * 0.1).astype(np.float32))
    S_n = 4
    act = torch.ones(1, 2 * nc, S_n, dtype=torch.bool)
    act[:, :nc] = False
    for s in range(S_n):
        act[0, nc + (5 + 11 * s) % nc, s] = False
    shifts = [[1, 0], [0, 2], [-1, -1], [2, 1]]
    h_mask = torch.zeros(1, cfgc.ctx_tokens, dtype=torch.bool)
    torch.manual_seed(6)
    with torch.no_grad():
        # sample_batch_size >= 2 * B * S makes the reference run all S prompts in one chunk (prediction.py:504-507)
        ys, fs = Gc.predict_counterfactual_videos_and_flows(imgc, active_patches=act.clone(), shifts=[list(v) for v in shifts], num_samples=S_n,
                                                            sample_batch_size=64, x_context=imu, mask_context=h_mask)
    np.savez_compressed(os.path.join(HERE, "wrapper_conj.npz"), img=imgc.numpy(), imu=imu.numpy(), active=act.numpy(),
                        shifts=np.array(shifts, dtype=np.int32), ys=ys.numpy(), flows=fs.numpy(), seed=np.array(5))
    print("[golden] wrapper_conj.npz", tuple(ys.shape), tuple(fs.shape))


def flowstats_inputs(seed=0, shape=(2, 2, 16, 16, 12)):
    """Seeded random flow samples [B,2,H,W,S] (one nearly-static sample, one constant sample)."""
    g = torch.Generator().manual_seed(seed)
    fl = torch.randn(*shape, generator=g) * 3
    fl[..., 3] *= 0.01
    fl[..., 5] = 0.25
    return fl


FLOW_OPTION_CASES = {
    "thresh_cov": dict(thresh=1.5, use_covariance=True),
    "thresh_binarize_corr": dict(thresh=1.5, binarize=True),
    "range_thresh_cov": dict(range_thresh=0.4, use_covariance=True),
    "normalize_cov": dict(normalize=True, use_covariance=True),
    "zscore_corr": dict(zscore=True),
    "spearman_corr": dict(do_spearman=True),
    "spearman_zscore_cov": dict(do_spearman=True, zscore=True, use_covariance=True),
    "thresh_normalize_zscore_cov": dict(thresh=0.5, normalize=True, zscore=True, use_covariance=True),
}


def run_flowstats_case(ns):
    """`compute_flow_corrs` / `compute_mean_motion_map` / `compute_flow_samples_magnitude` of the reference
    (cwm/models/segmentation.py:250-276, 479-547) on seeded random flows."""
    FG = ns.segmentation.FlowGenerator

    class _Self:  # the two map functions only use `self` to reach each other
        pass

    me = _Self()
    me.compute_flow_samples_magnitude = lambda *a, **k: FG.compute_flow_samples_magnitude(me, *a, **k)
    fl = flowstats_inputs()
    out = {"seed": 0, "shape": np.array(fl.shape)}
    for ds in (1, 2, 4):
        rows = slice(0, 2) if ds == 1 else slice(None)  # ds = 1: the first two image rows of source positions only (file size)
        out["cov_ds%d" % ds] = FG.compute_flow_corrs(fl, downsample=ds, use_covariance=True).numpy()[:, :, rows]
        out["corr_ds%d" % ds] = FG.compute_flow_corrs(fl, downsample=ds, use_covariance=False).numpy()[:, :, rows]
    for nps in (0, 1):
        for nm in (0, 1):
            out["map_nps%d_n%d" % (nps, nm)] = FG.compute_mean_motion_map(me, fl, normalize_per_sample=bool(nps), normalize=bool(nm)).numpy()
    out["mag_norm"] = FG.compute_flow_samples_magnitude(me, fl, normalize=True).numpy()
    out["map_4d"] = FG.compute_mean_motion_map(me, fl[..., 0].norm(dim=1, keepdim=True)).numpy()
    out["cov_one_sample"] = FG.compute_flow_corrs(fl[..., :1], downsample=2, use_covariance=True).numpy()
    np.savez_compressed(os.path.join(HERE, "flowstats.npz"), **out)
    print("flowstats.npz written")
    # round 4: the optional prologues (segmentation.py:519-538), one fixture per option and two combinations
    opt = {"seed": 0}
    for name, kw in FLOW_OPTION_CASES.items():
        opt[name] = FG.compute_flow_corrs(fl, downsample=2, **kw).numpy()
    np.savez_compressed(os.path.join(HERE, "flowstats_options.npz"), **opt)
    print("flowstats_options.npz written", {k: v.shape for k, v in opt.items() if k != "seed"})


def run_init_case(ns):
    """Reference constructor RNG parity: seed -> freshly initialised parameters (vmae.py:90,209,371)."""
    out = {}
    for seed in (0, 5):
        from functools import partial

        torch.manual_seed(seed)
        m = ns.vmae.PretrainVisionTransformer(
            img_size=32, patch_size=(8, 8), encoder_embed_dim=128, encoder_depth=2, encoder_num_heads=2,
            encoder_num_classes=0, decoder_embed_dim=128, decoder_depth=1, decoder_num_heads=2, mlp_ratio=4,
            qkv_bias=True, num_frames=2, tubelet_size=1, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
        sd = m.state_dict()
        out[f"keys_{seed}"] = np.array(list(sd.keys()))
        out[f"sums_{seed}"] = np.array([v.double().sum().item() for v in sd.values()])
        out[f"abs_{seed}"] = np.array([v.double().abs().sum().item() for v in sd.values()])
    np.savez_compressed(os.path.join(HERE, "init_tiny.npz"), **out)
    print("[golden] init_tiny.npz")


def run_sharp_cases(ns, skip_large=False):
    """Numerically hostile weights (`synthetic.sharpen_state_dict`: near one-hot softmax, LayerNorm weights U(0.2, 3), residual
    growth) through the reference: pins parity where operand rounding hurts most (VideoMAE/utils.py:87-121).  ViT-L/4 (36 blocks; the
    reference's own fp32 rounding there: 9.5e-6 against a float64 evaluation) and the IMU-conditioned model as well."""
    run_model_case(ns, TINY, batch=2, k_vis=4, clump=1, seed=6, out_name="tiny_8x8_sharp.npz", sharp=True)
    run_model_case(ns, C.CONFIGS["base_8x8patch_2frames_1tube"], batch=1, k_vis=8, clump=1, seed=2, out_name="base8_sharp_b1.npz", sharp=True)
    if skip_large:
        return
    run_model_case(ns, C.CONFIGS["large_4x4patch_2frames_1tube"], batch=1, k_vis=32, clump=2, seed=3, out_name="large4_sharp_b1.npz",
                   through_wrapper=False, sharp=True)
    cfg = C.CONJ_CONFIGS["imu400_base_4x4patch_2frames_1tube"]
    t0 = time.time()
    m = build_ref_conj(ns, cfg, 4, sharp=True)
    x = torch.from_numpy(S.synthetic_frames(1, cfg.main, 4))
    mask = torch.from_numpy(S.synthetic_masks(1, cfg.main, 4, 4))
    imu = torch.from_numpy((np.random.Generator(np.random.PCG64(11)).standard_normal((1, 6, 400)) * 0.1).astype(np.float32))
    mc = torch.zeros(1, 25, dtype=torch.bool)
    mc[0, 7] = True
    mean = torch.tensor(C.IMAGENET_MEAN).view(1, 3, 1, 1, 1)
    std = torch.tensor(C.IMAGENET_STD).view(1, 3, 1, 1, 1)
    with torch.no_grad():
        y = m((x.transpose(1, 2) - mean) / std, mask.clone(), x_context=imu, mask_context=mc)
    np.savez_compressed(os.path.join(HERE, "conj_imu400_sharp_b1.npz"), mask=mask.numpy(), imu=imu.numpy(), mask_context=mc.numpy(), y_tokens=y.numpy(),
                        seed=np.array(4), sharp=np.array(True))
    print(f"[golden] conj_imu400_sharp_b1.npz {tuple(y.shape)} std {y.std():.4f} ({time.time() - t0:.1f}s)")


TINY16 = C.VmaeConfig(name="tiny_16x16", img_size=(64, 64), patch=16, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128, dec_depth=1, dec_heads=2)


def run_round4_cases(ns, skip_large=False):
    """Round 4 (VERDICT r3 "next" item 4): the surface holes, every output produced by the reference.
    (a) P = 16: `base_16x16patch_2frames_1tube` (vmae.py:597-603) at full size and a tiny 16x16-patch model;
    (b) the conjoined model's context-stream output: `forward(..., output_context=True)` -> (x, x_c) and output_main=False -> x_c
        (conjoined_vmae.py:852-887, 990-1011), tiny (ragged visible counts, masked context tokens) and full size;
    (c) BASELINE configs[3] at full size: the reference's own prompt construction over all 256 synthetic prompts (one rectangularisation,
        torch seed 3) and its `predict` on three of them -- the first, the last and the first whose shift leaves the frame."""
    # ---- (a) ------------------------------------------------------------------------------------------------------------
    run_model_case(ns, TINY16, batch=2, k_vis=3, clump=1, seed=8, out_name="tiny_16x16_k3.npz")
    run_model_case(ns, C.CONFIGS["base_16x16patch_2frames_1tube"], batch=2, k_vis=8, clump=1, seed=0, out_name="base16_k8_b2.npz")

    # ---- (b) tiny: the inputs of conj_tiny.npz (B = 3, ragged visible counts, two rows with masked context tokens) -----------
    cfg = TINY_CONJ
    m = build_ref_conj(ns, cfg, 5)
    g = np.random.Generator(np.random.PCG64(2))
    B, n = 3, cfg.main.tokens_per_frame
    x = torch.from_numpy(g.random((B, 2, 3, 32, 32), dtype=np.float32))
    mask = torch.zeros(B, 2 * n, dtype=torch.bool)
    mask[:, n:] = True
    for b, vis in enumerate(([3, 9], [5], [6, 7, 16])):
        mask[b, [n + v for v in vis]] = False
    imu = torch.from_numpy((g.standard_normal((B, 6, cfg.ctx_seq_len)) * 0.1).astype(np.float32))
    mc = torch.zeros(B, cfg.ctx_tokens, dtype=torch.bool)
    mc[0, 1] = True
    mc[1, 0] = True
    mc[1, 3] = True
    G = ns.prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2, seed=0)
    out = {"x": x.numpy(), "mask": mask.numpy(), "imu": imu.numpy(), "mask_context": mc.numpy(), "seed": np.array(5)}
    with torch.no_grad():
        m._reset_padding_mask()
        y, y_c = m(G._preprocess(x), mask.clone(), x_context=imu, mask_context=mc, output_main=True, output_context=True)
        out["y_tokens"], out["y_ctx_tokens"] = y.numpy(), y_c.numpy()
        out["ctx_null_mask"] = m.context_stream.null_mask.numpy()
        m._reset_padding_mask()
        y_c2 = m(G._preprocess(x), mask.clone(), x_context=imu, mask_context=mc, output_main=False, output_context=True)
        assert torch.is_tensor(y_c2) and torch.equal(y_c2, y_c)
        m._reset_padding_mask()
        # the flags are sticky (`_set_decoder_outputs`, :589-593): a call without them still returns the context output alone
        y_c3 = m(G._preprocess(x), mask.clone(), x_context=imu, mask_context=mc)
        assert torch.is_tensor(y_c3) and torch.equal(y_c3, y_c)
        m._reset_padding_mask()
        both = m(G._preprocess(x), mask.clone(), x_context=imu, mask_context=mc, output_main=False, output_context=False)
        assert isinstance(both, tuple) and torch.equal(both[0], y) and torch.equal(both[1], y_c)   # "all the tokens from both streams"
    np.savez_compressed(os.path.join(HERE, "conj_tiny_ctx.npz"), **out)
    print("[golden] conj_tiny_ctx.npz", out["y_tokens"].shape, out["y_ctx_tokens"].shape)

    # ---- (c) 256 prompts on one frame pair: prompt construction by the reference, predict on three rows ------------------------
    class DummyFlow(torch.nn.Module):
        def forward(self, x, *a, **k):
            return torch.zeros(x.shape[0], x.shape[1] - 1, 2, *x.shape[-2:])

    cfg8 = C.CONFIGS["base_8x8patch_2frames_1tube"]
    t0 = time.time()
    mref = build_ref_model(ns, cfg8, 0)
    Gf = ns.segmentation.FlowGenerator(predictor=mref, flow_model=DummyFlow(), imagenet_normalize_inputs=True, temporal_dim=2, seed=0)
    table = S.synthetic_prompts(256, cfg8, 0)
    n8, gw = cfg8.tokens_per_frame, cfg8.img_size[1] // cfg8.patch
    x0 = torch.from_numpy(S.synthetic_frames(1, cfg8, 0))[:, 0:1]
    xs = x0.expand(-1, 2, -1, -1, -1).clone()          # make_static_movie: frame 1 := frame 0 (prediction.py:731-739)
    Sn = table.shape[0]
    active = torch.ones(1, 2 * n8, Sn, dtype=torch.bool)
    active[:, :n8] = False
    active[0, n8 + torch.from_numpy(table[:, 0].astype(np.int64)) * gw + torch.from_numpy(table[:, 1].astype(np.int64)), torch.arange(Sn)] = False
    passive = torch.ones(1, 2 * n8, Sn, dtype=torch.bool)
    passive[:, :n8] = False
    Gf.set_input(xs)
    Gf.shifter.set_shapes(xs, mask=active[..., 0])
    torch.manual_seed(3)
    x_shift, mask_post = Gf.create_motion_counterfactuals(xs, masks=passive, active_patches=active, shifts=[[int(v[2]), int(v[3])] for v in table],
                                                          num_samples=Sn, fix_passive=False, reset_shifts=True)
    dest = table[:, 0:2] + table[:, 2:4]
    oob = np.nonzero((dest < 0).any(1) | (dest[:, 0] >= cfg8.img_size[0] // cfg8.patch) | (dest[:, 1] >= gw))[0]
    rows = [0, int(oob[0]), Sn - 1]
    ys = []
    with torch.no_grad():
        for r in rows:
            ys.append(Gf.predict(x_shift[r:r + 1], mask_post[r:r + 1].clone(), frame=-1))
    ys = torch.cat(ys, 0)
    np.savez_compressed(os.path.join(HERE, "prompts256_rows.npz"), rows=np.array(rows), mask_post_rows=mask_post[rows].numpy(),
                        n_masked=mask_post.sum(-1).numpy().astype(np.int32), mask_post_digest=np.array([int(mask_post.sum()),
                        int((mask_post * torch.arange(mask_post.shape[1])[None]).sum())]), y_rows_even=ys[:, :, :, ::2].numpy().copy(),
                        seed=np.array(3))
    print(f"[golden] prompts256_rows.npz rows {rows} y {tuple(ys.shape)} ({time.time() - t0:.1f}s)")
    if skip_large:
        return
    # ---- (b) full size: B = 1, three masked context tokens ------------------------------------------------------------------
    cfgc = C.CONJ_CONFIGS["imu400_base_4x4patch_2frames_1tube"]
    t0 = time.time()
    m = build_ref_conj(ns, cfgc, 0)
    x = torch.from_numpy(S.synthetic_frames(1, cfgc.main, 6))
    mask = torch.from_numpy(S.synthetic_masks(1, cfgc.main, 4, 6))
    imu = torch.from_numpy((np.random.Generator(np.random.PCG64(13)).standard_normal((1, 6, 400)) * 0.1).astype(np.float32))
    mc = torch.zeros(1, 25, dtype=torch.bool)
    mc[0, [2, 11, 24]] = True
    mean = torch.tensor(C.IMAGENET_MEAN).view(1, 3, 1, 1, 1)
    std = torch.tensor(C.IMAGENET_STD).view(1, 3, 1, 1, 1)
    with torch.no_grad():
        y, y_c = m((x.transpose(1, 2) - mean) / std, mask.clone(), x_context=imu, mask_context=mc, output_main=True, output_context=True)
    np.savez_compressed(os.path.join(HERE, "conj_imu400_ctx_b1.npz"), mask=mask.numpy(), imu=imu.numpy(), mask_context=mc.numpy(),
                        y_ctx_tokens=y_c.numpy(), y_tokens_digest=np.array([y.double().sum().item(), y.double().abs().sum().item()]),
                        y_tokens_head=y[:, :8].numpy().copy(), seed=np.array(0), frames_seed=np.array(6))
    print(f"[golden] conj_imu400_ctx_b1.npz y_ctx {tuple(y_c.shape)} std {y_c.std():.4f} ({time.time() - t0:.1f}s)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-large", action="store_true")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    ns = ref_import.import_reference()
    if args.only == "init":
        run_init_case(ns)
        return
    if args.only == "shift":
        run_shift_cases(ns)
        return
    if args.only == "conj":
        run_conj_cases(ns, args.skip_large)
        return
    if args.only == "allvis":
        run_model_case(ns, TINY, batch=2, k_vis=16, clump=1, seed=5, out_name="tiny_8x8_allvis.npz", through_wrapper=False)
        return
    if args.only == "wrapper":
        run_wrapper_cases(ns)
        return
    if args.only == "flowstats":
        run_flowstats_case(ns)
        return
    if args.only == "sharp":
        run_sharp_cases(ns, args.skip_large)
        return
    if args.only == "r4":
        run_round4_cases(ns, args.skip_large)
        return
    run_init_case(ns)
    run_flowstats_case(ns)
    run_shift_cases(ns)
    run_conj_cases(ns, args.skip_large)
    run_wrapper_cases(ns)
    run_index_cases(ns)
    run_block_case(ns)
    run_model_case(ns, TINY, batch=3, k_vis=4, clump=1, seed=3, out_name="tiny_8x8_k4.npz")
    run_model_case(ns, TINY, batch=2, k_vis=1, clump=1, seed=4, out_name="tiny_8x8_k1.npz")
    # nothing masked: the decoder returns head(norm(x)) for ALL tokens (vmae.py:252-253); the wrapper cannot compose a video from it
    run_model_case(ns, TINY, batch=2, k_vis=16, clump=1, seed=5, out_name="tiny_8x8_allvis.npz", through_wrapper=False)
    base = C.CONFIGS["base_8x8patch_2frames_1tube"]
    run_model_case(ns, base, batch=2, k_vis=8, clump=1, seed=0, out_name="base8_k8_b2.npz")
    run_model_case(ns, base, batch=1, k_vis=1, clump=1, seed=1, out_name="base8_k1_b1.npz")
    run_sharp_cases(ns, args.skip_large)
    run_round4_cases(ns, args.skip_large)
    if not args.skip_large:
        large = C.CONFIGS["large_4x4patch_2frames_1tube"]
        run_model_case(ns, large, batch=1, k_vis=32, clump=2, seed=0, out_name="large4_k32_b1.npz",
                       through_wrapper=False)


if __name__ == "__main__":
    main()
