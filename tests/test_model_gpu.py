"""End-to-end parity of the HIP predictor (through the host mirror + C ABI) against the golden
reference outputs and the CPU oracle, plus size-independent properties at bench batch sizes."""
import os

import numpy as np
import pytest
import torch

from counterfactualworldmodels_amd import _lib, config as C, prediction, synthetic as S, vmae
from oracle import vmae_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TINY = C.VmaeConfig(name="tiny_8x8", img_size=(32, 32), patch=8, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128,
                    dec_depth=1, dec_heads=2)
TINY_SPEC = O.VmaeSpec(img_size=(32, 32), patch=8, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128, dec_depth=1, dec_heads=2)
TINY16 = C.VmaeConfig(name="tiny_16x16", img_size=(64, 64), patch=16, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128, dec_depth=1, dec_heads=2)
PARITY_TOL = 1e-3  # BASELINE.json north_star: outputs within 1e-3 max-abs of the CPU reference


def build(cfg, seed, mode="parity", sharp=False):
    m = vmae.PretrainVisionTransformer(cfg, mode=mode)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, seed, sharp=sharp).items()})
    return m.to("cuda:0").eval()


def case_inputs(g, cfg):
    seed, batch, k_vis, clump = int(g["seed"]), int(g["batch"]), int(g["k_vis"]), int(g["clump"])
    x = torch.from_numpy(S.synthetic_frames(batch, cfg, seed))
    mask = torch.from_numpy(S.synthetic_masks(batch, cfg, k_vis, seed, clump))
    return seed, x, mask


@pytest.mark.parametrize("name", ["tiny_8x8_k4.npz", "tiny_8x8_k1.npz"])
def test_tiny_golden(name):
    g = np.load(os.path.join(GOLDEN, name))
    seed, x, mask = case_inputs(g, TINY)
    m = build(TINY, seed)
    G = prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
    xd, md = x.cuda(), mask.cuda()
    y = m(G._preprocess(xd), md).cpu()           # the reference seam (vmae.py:539)
    assert y.shape == g["y_tokens"].shape
    assert np.abs(y.numpy() - g["y_tokens"]).max() <= 2e-4
    video = G.predict(xd, md.clone(), frame=None).cpu()   # fused path
    rows = video[:, 1, :, :: max(1, TINY.img_size[0] // 8)].numpy()
    assert np.abs(rows - g["video_frame1_rows"]).max() <= 2e-4
    v = video.double()
    dig = np.array([v.sum().item(), v.abs().sum().item(), (v ** 2).sum().item()])
    assert np.allclose(dig, g["video_digest"], rtol=1e-5)
    # visible patches are bit-exact copies of the raw input
    with torch.no_grad():
        ref_video = O.predict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(TINY, seed).items()}, TINY_SPEC, x, mask,
                              frame=None)
    vis_pix = O.patches_to_video((~mask)[..., None].expand(-1, -1, 192).float(), 2, 3, 32, 32, 8).bool()
    assert torch.equal(video[vis_pix], x[vis_pix])
    assert (video - ref_video).abs().max().item() <= 2e-4
    # frame=-1 returns the last frame only (prediction.py:447-449)
    last = G.predict(xd, md.clone()).cpu()
    assert last.shape == (x.shape[0], 1, 3, 32, 32) and torch.equal(last[:, 0], video[:, 1])


@pytest.mark.parametrize("name", ["base8_k8_b2.npz", "base8_k1_b1.npz"])
def test_base8_golden_parity_and_fast(name):
    g = np.load(os.path.join(GOLDEN, name))
    cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
    seed, x, mask = case_inputs(g, cfg)
    m = build(cfg, seed, "parity")
    G = prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
    y = G.predict_tokens(x.cuda(), mask.cuda()).cpu().numpy()
    err = np.abs(y - g["y_tokens"]).max()
    print(f"[{name}] parity-mode max-abs vs reference: {err:.3e}")
    assert err <= PARITY_TOL, err
    video = G.predict(x.cuda(), mask.cuda(), frame=None).cpu()
    rows = video[:, 1, :, :: cfg.img_size[0] // 8].numpy()
    assert np.abs(rows - g["video_frame1_rows"]).max() <= PARITY_TOL
    m.mode = "fast"
    yf = G.predict_tokens(x.cuda(), mask.cuda()).cpu().numpy()
    errf = np.abs(yf - g["y_tokens"]).max()
    print(f"[{name}] fast-mode (plain bf16) max-abs vs reference: {errf:.3e}")
    # plain bf16 does not meet PARITY_TOL; its floor is 2x the measured error (2.5e-2 / 2.6e-2 on these goldens), so that the `secondary` bench
    # line cannot silently get worse
    assert errf <= 5e-2 and np.abs(yf - g["y_tokens"]).mean() <= 1.5e-2


@pytest.mark.parametrize("name,cfg_name", [("tiny_16x16_k3.npz", None), ("base16_k8_b2.npz", "base_16x16patch_2frames_1tube")])
def test_patch16_goldens(name, cfg_name):
    """P = 16 on the HIP path (vmae.py:597-603 `base_16x16patch_2frames_1tube`: 392 tokens, patch-embed K = 768, head N = 768; and a tiny
    16x16-patch model) against the reference's outputs: tokens through the model seam, frames through the fused `predict`."""
    g = np.load(os.path.join(GOLDEN, name))
    cfg = TINY16 if cfg_name is None else C.CONFIGS[cfg_name]
    seed, x, mask = case_inputs(g, cfg)
    for mode, tol in (("parity", 2e-4 if cfg_name is None else PARITY_TOL), ("fast", 6e-2)):
        if cfg_name is None:
            m = build(cfg, seed, mode=mode)
        else:
            m = vmae.base_16x16patch_2frames_1tube(mode=mode)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, seed).items()})
            m = m.to("cuda:0").eval()
            assert m.patch_size == (1, 16, 16) and m.mask_size == (2, 14, 14)
        G = prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
        y = m(G._preprocess(x.cuda()), mask.cuda()).cpu().numpy()
        err = np.abs(y - g["y_tokens"]).max()
        print(f"[{name} {mode}] max-abs vs reference {err:.2e}")
        assert y.shape == g["y_tokens"].shape and err <= tol, (mode, err)
        video = G.predict(x.cuda(), mask.cuda().clone(), frame=None).cpu()
        rows = video[:, 1, :, :: max(1, cfg.img_size[0] // 8)].numpy()
        assert np.abs(rows - g["video_frame1_rows"]).max() <= tol
        if mode == "parity":
            v = video.double()
            assert np.allclose([v.sum().item(), v.abs().sum().item(), (v ** 2).sum().item()], g["video_digest"], rtol=1e-5)


def test_checkpoint_file_to_hip_path(tmp_path):
    """f-3 end to end (prediction.py:81-107): a state-dict saved as `{"model": ...}` .pth, loaded through `predictor_load_path=` into a
    predictor that held OTHER weights, must reach the packed weights of the library: HIP output = oracle output on the saved weights."""
    sd = {k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(TINY, 11).items()}
    path = str(tmp_path / "tiny_ckpt.pth")
    torch.save({"model": sd, "epoch": 3}, path)
    m = build(TINY, 12)                                                     # other weights, already packed by a forward
    x = torch.from_numpy(S.synthetic_frames(2, TINY, 13))
    mask = torch.from_numpy(S.synthetic_masks(2, TINY, 4, 13))
    G0 = prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
    before = G0.predict(x.cuda(), mask.cuda().clone(), frame=None).cpu()
    G = prediction.PredictorBasedGenerator(predictor=m, predictor_load_path=path, imagenet_normalize_inputs=True, temporal_dim=2)
    assert G._predictor_load_path == path
    with torch.no_grad():
        ref = O.predict(sd, TINY_SPEC, x, mask, frame=None)
    out = G.predict(x.cuda(), mask.cuda().clone(), frame=None).cpu()
    assert (out - ref).abs().max().item() <= 2e-4
    assert (before - ref).abs().max().item() > 1e-2                         # the file's weights differ from the ones it replaced
    # the plain state-dict form (no "model" key) and load_predictor on an existing wrapper
    torch.save({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(TINY, 12).items()}, path)
    G.load_predictor(path)
    back = G.predict(x.cuda(), mask.cuda().clone(), frame=None).cpu()
    assert (back - before).abs().max().item() <= 1e-6


@pytest.mark.parametrize("name", ["tiny_8x8_sharp.npz", "base8_sharp_b1.npz"])
def test_sharp_weights_golden_parity(name):
    """Numerically hostile weights (`synthetic.sharpen_state_dict`: sharp softmax with logits up to +-21, LayerNorm weights U(0.2, 3),
    residual growth x2), run through the REFERENCE by make_golden.py.  Parity mode must hold the 1e-3 tolerance here too; plain bf16
    (fast mode) is reported.  The setting is the sharpest at which the reference is itself reproducible in fp32 (1.3e-5)."""
    g = np.load(os.path.join(GOLDEN, name))
    cfg = TINY if name.startswith("tiny") else C.CONFIGS["base_8x8patch_2frames_1tube"]
    seed, x, mask = case_inputs(g, cfg)
    m = build(cfg, seed, "parity", sharp=True)
    G = prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
    y = G.predict_tokens(x.cuda(), mask.cuda()).cpu().numpy()
    err = np.abs(y - g["y_tokens"]).max()
    print(f"[{name}] sharp weights, parity-mode max-abs vs reference: {err:.3e}")
    assert err <= PARITY_TOL, err
    m.mode = "fast"
    yf = G.predict_tokens(x.cuda(), mask.cuda()).cpu().numpy()
    print(f"[{name}] sharp weights, fast-mode max-abs vs reference: {np.abs(yf - g['y_tokens']).max():.3e}")
    assert np.isfinite(yf).all() and np.abs(yf - g["y_tokens"]).max() <= 2.7e-1   # 2x the measured 1.3e-1


def test_sharp_weights_large4_parity():
    """ViT-L/4 (36 blocks) under the hostile weights, against the reference's output: parity mode inside 1e-3 (the CPU emulation of
    the split-bf16 arithmetic predicts 1.6e-4)."""
    g = np.load(os.path.join(GOLDEN, "large4_sharp_b1.npz"))
    cfg = C.CONFIGS["large_4x4patch_2frames_1tube"]
    seed = int(g["seed"])
    x = torch.from_numpy(S.synthetic_frames(1, cfg, seed))
    mask = torch.from_numpy(S.synthetic_masks(1, cfg, 32, seed, 2))
    m = build(cfg, seed, "parity", sharp=True)
    G = prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
    y = G.predict_tokens(x.cuda(), mask.cuda()).cpu().numpy()
    err = np.abs(y - g["y_tokens"]).max()
    print(f"[large4 sharp] parity-mode max-abs vs reference: {err:.3e}")
    assert err <= PARITY_TOL, err


def test_large4_golden_parity():
    g = np.load(os.path.join(GOLDEN, "large4_k32_b1.npz"))
    cfg = C.CONFIGS["large_4x4patch_2frames_1tube"]
    seed = int(g["seed"])
    x = torch.from_numpy(S.synthetic_frames(1, cfg, seed))
    mask = torch.from_numpy(S.synthetic_masks(1, cfg, 32, seed, 2))
    m = build(cfg, seed, "parity")
    G = prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
    y = G.predict_tokens(x.cuda(), mask.cuda()).cpu().numpy()
    err = np.abs(y - g["y_tokens"]).max()
    print(f"[large4] parity-mode max-abs vs reference: {err:.3e}")
    assert err <= PARITY_TOL, err
    m.mode = "fast"
    yf = G.predict_tokens(x.cuda(), mask.cuda()).cpu().numpy()
    print(f"[large4] fast-mode max-abs vs reference: {np.abs(yf - g['y_tokens']).max():.3e}")
    assert np.abs(yf - g["y_tokens"]).max() <= 7e-2   # 2x the measured 3.5e-2


def test_bench_batch_properties():
    """At BASELINE.json's full size (B/8, batch 32) the oracle takes minutes, so check
    size-independent properties: per-sample independence (a row's output does not depend on its
    batch-mates or its position), determinism, and agreement of rows 0-1 with the golden B=2 case."""
    g = np.load(os.path.join(GOLDEN, "base8_k8_b2.npz"))
    cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
    m = build(cfg, 0, "parity")
    B = 32
    x = torch.from_numpy(S.synthetic_frames(B, cfg, 0)).cuda()
    mask = torch.from_numpy(S.synthetic_masks(B, cfg, 8, 0)).cuda()
    xp = O.preprocess(x.cpu()).cuda()
    y = m(xp, mask, n_vis=792)
    assert torch.isfinite(y).all()
    # the first two rows of the B=32 synthetic batch are exactly the golden B=2 inputs
    assert np.array_equal(mask[:2].cpu().numpy(), g["mask"])
    assert np.abs(y[:2].cpu().numpy() - g["y_tokens"]).max() <= PARITY_TOL
    for rep in range(30):  # deterministic, run after run (a timing-dependent fault shows up as a rare difference: DESIGN.md 4.9 (5d))
        assert torch.equal(y, m(xp, mask, n_vis=792)), rep
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()
    yp = m(xp[perm], mask[perm], n_vis=792)
    assert (yp - y[perm]).abs().max().item() <= 1e-5  # batch-permutation equivariance
    ys = m(xp[5:6], mask[5:6], n_vis=792)
    # batch-size invariance (batch 1 takes other GEMM tilings and splits the long-K GEMMs over idle CUs: fp32 sums re-associate)
    assert (ys - y[5:6]).abs().max().item() <= 5e-5


def test_error_behaviour():
    m = build(TINY, 3)
    x = torch.zeros(2, 3, 2, 32, 32, device="cuda")
    ragged = torch.zeros(2, 32, dtype=torch.bool, device="cuda")
    ragged[0, 16:] = True
    ragged[1, 17:] = True
    with pytest.raises(RuntimeError):  # the reference's reshape fails on unequal visible counts (vmae.py:167)
        m(x, ragged)
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 3, 2, 40, 40, device="cuda"), ragged)
    with pytest.raises(RuntimeError):
        m(x.cpu(), ragged.cpu())
    sd = m.state_dict()
    sd["bogus.weight"] = torch.zeros(3)
    with pytest.raises(RuntimeError):
        m.load_state_dict(sd)


def test_weight_reload_is_picked_up():
    m = build(TINY, 3)
    x = torch.from_numpy(S.synthetic_frames(2, TINY, 3)).cuda()
    mask = torch.from_numpy(S.synthetic_masks(2, TINY, 4, 3)).cuda()
    xp = O.preprocess(x.cpu()).cuda()
    y1 = m(xp, mask)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(TINY, 4).items()})
    y2 = m(xp, mask)
    W = {k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(TINY, 4).items()}
    with torch.no_grad():
        ref = O.vmae_forward(W, TINY_SPEC, xp.cpu(), mask.cpu())
    assert (y2.cpu() - ref).abs().max().item() <= 2e-4
    assert (y1 - y2).abs().max().item() > 1e-2


def test_weight_edits_are_detected_without_walking_the_state_dict():
    """`sync_weights` compares parameter version counters (vmae.WeightSync): optimiser-style in-place updates, `load_state_dict`, `.to()`
    and `sync_weights(force=True)` after a `.data` edit must all reach the library; an unchanged model uploads nothing."""
    m = build(TINY, 3)
    x = torch.from_numpy(S.synthetic_frames(2, TINY, 3)).cuda()
    mask = torch.from_numpy(S.synthetic_masks(2, TINY, 4, 3)).cuda()
    xp = O.preprocess(x.cpu()).cuda()
    y0 = m(xp, mask)
    assert m.sync_weights() == 0 and m._params_unchanged()
    with torch.no_grad():
        m.decoder.head.bias.add_(0.5)                      # in-place update: bumps the version counter
    y1 = m(xp, mask)
    assert ((y1 - y0) - 0.5).abs().max().item() <= 1e-5   # the head bias shifts every output by exactly 0.5
    m.decoder.head.bias.data.sub_(0.5)                     # `.data` edit: invisible to the counters ...
    assert torch.equal(m(xp, mask), y1)
    assert m.sync_weights(force=True) > 0                  # ... until the caller says so
    assert (m(xp, mask) - y0).abs().max().item() <= 1e-6
    m2 = m.cpu().cuda()                                    # `.to()` replaces the parameter tensors: the list is rebuilt
    assert (m2(xp, mask) - y0).abs().max().item() <= 1e-6
    # round-2 review: new storage behind the same version counter, a submodule-only conversion, a replaced Parameter
    m2.decoder.head.bias.data = m2.decoder.head.bias.data + 0.25
    assert ((m2(xp, mask) - y0) - 0.25).abs().max().item() <= 1e-5
    m2.decoder.double()
    assert ((m2(xp, mask) - y0) - 0.25).abs().max().item() <= 1e-5 and m2.sync_weights() == 0
    m2.decoder.float()
    m2.decoder.head.bias = torch.nn.Parameter(m2.decoder.head.bias.detach() - 0.25)
    assert (m2(xp, mask) - y0).abs().max().item() <= 1e-5


def test_qkv_epilogue_variants_agree_end_to_end():
    """Direct, LDS-staged and per-fragment GEMM epilogues (QKV head scatter included) and every tile configuration: same forward output,
    bit for bit (all of them apply the same product sequence to every accumulator).  The options are per model handle (cwm_model_set_option)."""
    g = np.load(os.path.join(GOLDEN, "tiny_8x8_k4.npz"))
    seed, x, mask = case_inputs(g, TINY)
    outs = {}
    for mode in ("parity", "fast"):
        m = build(TINY, seed, mode=mode)
        G = prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
        for staged, direct in ((1, 1), (1, 0), (0, 0), (1, 2)):
            for tile in (0, 1, 4, 6):
                m.set_option("gemm_staged", staged)
                m.set_option("gemm_direct", direct)
                m.set_option("gemm_tile", tile)
                outs[(mode, staged * 10 + direct, tile)] = m(G._preprocess(x.cuda()), mask.cuda()).cpu()
        ref = outs[(mode, 11, 0)]
        for key, y in outs.items():
            if key[0] == mode:
                assert torch.equal(y, ref), key
    assert np.abs(outs[("parity", 11, 0)].numpy() - g["y_tokens"]).max() <= 2e-4


def test_fused_index_prologue_is_bitwise_equal_to_the_four_launches():
    """Round 5: mask -> permutation, its inverse, the per-row visible-count check and the patch gather run as ONE launch (index_gather_kernel)
    instead of memset + mask_to_perm + patch_gather + perm_to_rank.  Same tokens, same video, bit for bit, for the tiny model and ViT-B/8 (one and two
    lanes), with the un-embed (which reads the inverse permutation) and without; a row with the wrong visible count is still refused, whichever row of
    whichever lane it is in, and a later good call on the same handle is unaffected (the row words are rewritten by every call: no memset)."""
    for cfg, name in ((TINY, "tiny_8x8_k4.npz"), (C.CONFIGS["base_8x8patch_2frames_1tube"], "base8_k8_b2.npz")):
        g = np.load(os.path.join(GOLDEN, name))
        seed, x, mask = case_inputs(g, cfg)
        reps = 5 if cfg is TINY else 8   # ViT-B/8: batch 16 = two lanes
        xb, mb = x.repeat(reps, 1, 1, 1, 1).cuda(), mask.repeat(reps, 1).cuda()
        m = build(cfg, seed)
        outs = []
        for fused in (1, 0):
            m.set_option("index_fused", fused)
            tok, vid = m.predict_video(xb, mb)
            outs.append((tok.clone(), vid.clone(), m(prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)._preprocess(xb), mb).clone()))
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b), name
        m.set_option("index_fused", 1)
        for bad_row in (0, xb.shape[0] - 1):
            bad = mb.clone()
            col = int(torch.nonzero(bad[bad_row])[0])
            bad[bad_row, col] = False   # one more visible token in one row
            with pytest.raises(_lib.CwmHipError) as ei:
                m.predict_video(xb, bad)
            assert ei.value.code == _lib.ERR_MASK
            tok, vid = m.predict_video(xb, mb)
            assert torch.equal(tok, outs[0][0]) and torch.equal(vid, outs[0][1])


def test_options_are_per_model_handle():
    """Two models in one process with different execution options do not see each other's settings (until round 4 the switches were process-wide
    globals).  Model A is forced onto the 256x256 8-phase GEMM kernel and the 4-wave attention kernel, model B keeps the defaults (which never pick
    the 8-phase kernel for this small model); called alternately, the per-handle kernel timers show A's launches on the wide kernel and none of B's,
    both reproduce their own results bit for bit, and an unknown option is refused."""
    g = np.load(os.path.join(GOLDEN, "tiny_8x8_k4.npz"))
    seed, x, mask = case_inputs(g, TINY)
    a, b = build(TINY, seed), build(TINY, seed)
    G = prediction.PredictorBasedGenerator(predictor=a, imagenet_normalize_inputs=True, temporal_dim=2)
    xp, mk = G._preprocess(x.cuda()), mask.cuda()
    a.set_option("attn_kernel", 1)   # before the first forward: applied when the handle is created
    ya, yb = a(xp, mk), b(xp, mk)
    a.set_option("gemm_tile", 4)     # on the live handle
    for m in (a, b):
        m.timing_enable(_lib.KCLASS_GEMM, True)
    for _ in range(3):
        assert torch.equal(a(xp, mk), ya)   # (every tile configuration is bit-identical: test above)
        assert torch.equal(b(xp, mk), yb)
    wide = {}
    for name, m in (("a", a), ("b", b)):
        m.timing_collect(_lib.KCLASS_GEMM)
        wide[name] = (m.timing_collect(_lib.KCLASS_GEMM_WIDE)["launches"], m.timing_collect(_lib.KCLASS_GEMM_NARROW)["launches"])
        m.timing_enable(_lib.KCLASS_GEMM, False)
    assert wide["a"][0] > 0 and wide["a"][1] == 0, wide
    assert wide["b"][0] == 0 and wide["b"][1] > 0, wide
    assert (ya - yb).abs().max().item() <= 5e-5
    with pytest.raises(RuntimeError):
        b.set_option("no_such_option", 1)
    assert "no_such_option" not in b.__dict__.get("_options", {}) and torch.equal(b(xp, mk), yb)
    # a refused option leaves nothing behind: the handle can be re-created (device move / use_library replay the stored options)
    b._release()
    assert torch.equal(b(xp, mk), yb)
    # ... also when there is no handle yet (the key list is checked on the host)
    c = build(TINY, seed)
    c._release()
    with pytest.raises(RuntimeError):
        c.set_option("no_such_option", 1)
    with pytest.raises(RuntimeError, match="timing-only"):
        c.set_option("gemm_debug", 8)
    assert not c.__dict__.get("_options") and torch.equal(c(xp, mk), yb)
    # the production setters refuse the timing-only ablation bits (they would return wrong outputs with CWM_OK); the result-preserving bits pass
    with pytest.raises(RuntimeError, match="timing-only"):
        c.set_option("gemm_debug", 2)
    c.set_option("gemm_debug", 32)
    assert (c(xp, mk) - yb).abs().max().item() <= 5e-5


def test_forward_from_another_current_device_is_refused():
    """A handle belongs to the device it was created on (weights, workspace, lane streams).  `cwm_forward` / `cwm_*_load_weight` called while another
    device is current return CWM_ERR_INVALID instead of launching on the wrong GPU.  One GPU here: the development library's "pretend_device" makes this
    thread's checks see device 1 as current (the handle lives on device 0)."""
    g = np.load(os.path.join(GOLDEN, "tiny_8x8_k4.npz"))
    seed, x, mask = case_inputs(g, TINY)
    dlib = _lib.get_dev_lib()
    m = build(TINY, seed)
    m.use_library(dlib)
    G = prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
    xp, mk = G._preprocess(x.cuda()), mask.cuda()
    y = m(xp, mk)
    _lib.check(dlib.cwm_debug_set(b"pretend_device", 1), dlib)
    try:
        with pytest.raises(RuntimeError, match="created on HIP device 0 but the calling thread's current device is 1"):
            m(xp, mk)
        with pytest.raises(RuntimeError, match="current device is 1"):
            m.sync_weights(torch.device("cuda:0"), force=True)
    finally:
        _lib.check(dlib.cwm_debug_set(b"pretend_device", -1), dlib)
    m.sync_weights(torch.device("cuda:0"), force=True)
    assert torch.equal(m(xp, mk), y)


def test_nothing_masked_returns_all_tokens():
    """Edge case of the reference decoder (vmae.py:250-253): no masked token -> head(norm(x)) of all Nt tokens; the fused
    video path refuses, like the reference wrapper's failing assignment (prediction.py:252-254)."""
    g = np.load(os.path.join(GOLDEN, "tiny_8x8_allvis.npz"))
    seed = int(g["seed"])
    m = build(TINY, seed)
    G = prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
    x, mask = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["mask"]).cuda()
    y = m(G._preprocess(x), mask).cpu().numpy()
    assert y.shape == g["y_tokens"].shape
    assert np.abs(y - g["y_tokens"]).max() <= 2e-4
    with pytest.raises(RuntimeError):
        m.predict_video(x, mask)


def test_last_decoder_block_pruning_is_exact():
    """The last decoder block computes queries / proj / MLP only for the Nm rows the head reads (vmae.py:250-251 discards the rest):
    outputs are bit-identical to running the block in full, for the tiny model and for B/8 (both modes).  (Bitwise without split-K --
    the pruned GEMMs have fewer rows, so the small-launch heuristic may split K differently and re-associate the fp32 sums -- and with
    every attention tile on the regular schedule: pruning changes WHICH query rows form the ragged last tile, whose key-split
    schedule (attention_tail.h) rounds P at other values.  In the default configuration the two agree to rounding.)"""
    cases = [(TINY, "tiny_8x8_k4.npz"), (C.CONFIGS["base_8x8patch_2frames_1tube"], "base8_k8_b2.npz")]
    for cfg, name in cases:
        g = np.load(os.path.join(GOLDEN, name))
        seed, x, mask = case_inputs(g, cfg)
        for mode in ("parity", "fast"):
            m = build(cfg, seed, mode)
            for k, v in (("gemm_debug", 32), ("attn_tail", 0), ("attn_ksplit", 0)):
                m.set_option(k, v)
            G = prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
            outs = []
            for prune in (1, 0):
                m.set_option("prune_last_block", prune)
                outs.append(m(G._preprocess(x.cuda()), mask.cuda()).cpu())
            assert torch.equal(outs[0], outs[1]), (name, mode, (outs[0] - outs[1]).abs().max().item())
    for cfg, name in cases:  # library defaults (split-K where the heuristic takes it, key-split attention tails): equal to rounding
        g = np.load(os.path.join(GOLDEN, name))
        seed, x, mask = case_inputs(g, cfg)
        m = build(cfg, seed, "parity")
        G = prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
        outs = []
        for prune in (1, 0):
            m.set_option("prune_last_block", prune)
            outs.append(m(G._preprocess(x.cuda()), mask.cuda()).cpu())
        assert (outs[0] - outs[1]).abs().max().item() <= 5e-5


def test_two_lanes_match_one_lane_and_report_mask_errors_of_both():
    """cwm_model_set_lanes: a batch of 17 runs as 9 + 8 on two streams (fork / join inside cwm_forward).  Tokens and video must be
    those of the single-lane call up to the per-shape kernel choice, the caller's stream must see finished outputs, and a bad mask row
    in the SECOND lane must still raise the reference's error."""
    cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
    m = build(cfg, 3)
    B = 17
    x = torch.from_numpy(S.synthetic_frames(B, cfg, 5)).cuda()
    mask = torch.from_numpy(S.synthetic_masks(B, cfg, 8, 5)).cuda()
    n_vis = cfg.tokens_per_frame + 8
    y2, v2 = m.predict_video(x, mask, n_vis=n_vis)          # library default: two lanes
    m.set_lanes(1)
    y1, v1 = m.predict_video(x, mask, n_vis=n_vis)
    m.set_lanes(2)
    y2b, v2b = m.predict_video(x, mask, n_vis=n_vis)
    assert torch.equal(y2, y2b) and torch.equal(v2, v2b)    # deterministic
    err_t, err_v = (y2 - y1).abs().max().item(), (v2 - v1).abs().max().item()
    print(f"[lanes] 2 vs 1 lanes: tokens {err_t:.2e} video {err_v:.2e}")
    assert err_t <= 5e-5 and err_v <= 5e-5
    bad = mask.clone()
    row = B - 2                                              # second lane
    j = int(torch.nonzero(~bad[row])[0])
    bad[row, j] = True                                       # one visible token fewer in that row
    with pytest.raises(RuntimeError, match="is invalid for"):
        m.predict_video(x, bad, n_vis=n_vis)
    y3, _ = m.predict_video(x, mask, n_vis=n_vis)            # the model is usable afterwards
    assert torch.equal(y3, y2)
    # the reference-shaped entry (pre-processed [B,C,T,H,W] in, tokens out) takes the same two-lane path
    xp = O.preprocess(x.cpu()).cuda()
    yf2 = m(xp, mask, n_vis=n_vis)
    m.set_lanes(1)
    yf1 = m(xp, mask, n_vis=n_vis)
    m.set_lanes(2)
    assert (yf2 - yf1).abs().max().item() <= 5e-5 and (yf2 - y2).abs().max().item() <= 5e-5
    # work queued behind the call on the caller's stream sees both halves finished (the join is stream-ordered, not a host sync)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ys, _ = m.predict_video(x, mask, n_vis=n_vis, check=False)
        tail = ys[B - 1].clone()                               # last row = end of the second lane
    s.synchronize()
    assert torch.equal(tail, y2[B - 1])


@pytest.mark.parametrize("lanes", [3, 4])
def test_three_and_four_lanes_batch32(lanes):
    """cwm_model_set_lanes accepts 1 .. 4 (include/cwm_hip.h): every accepted value runs.  The bench batch (ViT-B/8, 32 frame pairs) on 3 lanes (11 + 11 + 10
    rows) and 4 lanes (8 each): rows 0-1 against the reference's golden pair, bit-stable run to run, <= 5e-5 from the one-lane call (per-shape kernel choice
    re-associates fp32 sums), the fused video path too, a bad mask row in EVERY lane raises the reference's error, and the handle falls back to fewer
    lanes by itself when a batch is too small to keep 3000 encoder rows per lane."""
    g = np.load(os.path.join(GOLDEN, "base8_k8_b2.npz"))
    cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
    m = build(cfg, 0, "parity")
    B, n_vis = 32, 792
    xr = torch.from_numpy(S.synthetic_frames(B, cfg, 0)).cuda()
    x = O.preprocess(xr.cpu()).cuda()
    mask = torch.from_numpy(S.synthetic_masks(B, cfg, 8, 0)).cuda()
    m.sync_weights(torch.device("cuda:0"))
    m.set_lanes(1)
    y1 = m(x, mask, n_vis=n_vis)
    m.set_lanes(lanes)
    yl = m(x, mask, n_vis=n_vis)
    err = np.abs(yl[:2].cpu().numpy() - g["y_tokens"]).max()
    e1 = (yl - y1).abs().max().item()
    print(f"[lanes={lanes}] rows 0-1 vs reference {err:.2e}; vs one lane {e1:.2e}")
    assert err <= PARITY_TOL and e1 <= 5e-5
    for rep in range(5):
        assert torch.equal(m(x, mask, n_vis=n_vis), yl), rep
    yv, video = m.predict_video(xr, mask, n_vis=n_vis)
    m.set_lanes(1)
    yv1, video1 = m.predict_video(xr, mask, n_vis=n_vis)
    m.set_lanes(lanes)
    assert (yv - yl).abs().max().item() <= 5e-5 and (video - video1).abs().max().item() <= 5e-5
    first = [(B * l + lanes - 1) // lanes for l in range(lanes + 1)]       # lane l owns rows [first[l], first[l+1]) (model.hip cwm_forward)
    for l in range(lanes):
        bad = mask.clone()
        row = first[l + 1] - 1
        bad[row, int(torch.nonzero(~bad[row])[0])] = True                  # one visible token fewer in the last row of lane l
        with pytest.raises(RuntimeError, match="is invalid for"):
            m(x, bad, n_vis=n_vis)
    assert torch.equal(m(x, mask, n_vis=n_vis), yl)                        # usable afterwards
    ys = m(x[:9], mask[:9], n_vis=n_vis)                                   # 9 rows: two lanes at most (3 x 3 rows would leave 2376 < 3000 encoder rows per lane)
    assert (ys - yl[:9]).abs().max().item() <= 5e-5
    with pytest.raises(RuntimeError):
        m.set_lanes(5)
    with pytest.raises(RuntimeError):
        m.set_lanes(0)


@pytest.mark.parametrize("B", [8, 9, 13])
def test_mid_batches_run_two_lanes(B):
    """Round 4 lowered the lane split to 3000 encoder rows per half (engine.h kMinLaneRows: ViT-B/8 from batch 8) and lets partly filled rounds inside a
    two-lane call take the 8-phase kernel: the mid batches therefore run other kernels than both batch 32 and batch 1.  Two lanes against one lane
    (<= 5e-5: kernel choice and split-K re-associate fp32 sums), rows 0-1 against the reference's golden pair, and bit-stable over repetitions."""
    g = np.load(os.path.join(GOLDEN, "base8_k8_b2.npz"))
    cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
    m = build(cfg, 0, "parity")
    x = O.preprocess(torch.from_numpy(S.synthetic_frames(B, cfg, 0))).cuda()
    mask = torch.from_numpy(S.synthetic_masks(B, cfg, 8, 0)).cuda()
    y2 = m(x, mask, n_vis=792)
    assert np.abs(y2[:2].cpu().numpy() - g["y_tokens"]).max() <= PARITY_TOL
    for rep in range(10):
        assert torch.equal(m(x, mask, n_vis=792), y2), rep
    m.set_lanes(1)
    y1 = m(x, mask, n_vis=792)
    m.set_lanes(2)
    assert (y2 - y1).abs().max().item() <= 5e-5
