"""BASELINE configs[4]: the IMU-conditioned conjoined padded predictor on the GPU (through the host mirror and
the C ABI) against the reference's golden outputs and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from counterfactualworldmodels_amd import config as C, conjoined_vmae as CV, prediction, synthetic as S
from oracle import conj_oracle as CO, vmae_oracle as V
from test_conj_oracle import TINY_CONJ, TINY_SPEC, conj_weights

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build(cfg, seed, mode="parity", sharp=False):
    m = CV.ConjoinedPaddedVisionTransformer(cfg, mode=mode)
    m.load_state_dict(conj_weights(cfg, seed, sharp))
    return m.cuda().eval()


def test_tiny_conj_golden_ragged_and_wrapper():
    g = np.load(os.path.join(GOLDEN, "conj_tiny.npz"))
    m = build(TINY_CONJ, int(g["seed"]))
    G = prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
    x, mask, imu, mc = (torch.from_numpy(g[k]).cuda() for k in ("x", "mask", "imu", "mask_context"))
    y = m(G._preprocess(x), mask, x_context=imu, mask_context=mc).cpu().numpy()
    assert y.shape == g["y_tokens"].shape
    assert np.abs(y - g["y_tokens"]).max() <= 3e-4
    assert hasattr(m, "padding_mask")
    # the padding state of both streams equals what the reference model leaves behind (conj_padding.npz; conjoined_vmae.py:49-116)
    pad = np.load(os.path.join(GOLDEN, "conj_padding.npz"))
    assert np.array_equal(m.main_stream.padding_mask.cpu().numpy(), pad["padding_mask"])
    assert np.array_equal(m.main_stream.full_input_mask.cpu().numpy(), pad["full_input_mask"])
    assert np.array_equal(m.main_stream.null_mask.cpu().numpy(), pad["null_mask"])
    assert np.array_equal(m.context_stream.padding_mask.cpu().numpy(), pad["ctx_padding_mask"])
    G.reset_padding_masks()
    assert not hasattr(m, "padding_mask") and m.context_stream.padding_mask is None
    # zero rows exactly where the reference has them (masked pad slots)
    assert np.array_equal(np.abs(y).sum(-1) == 0, np.abs(g["y_tokens"]).sum(-1) == 0)
    mask_eq = torch.from_numpy(g["mask_eq"]).cuda()
    mc_eq = torch.zeros(2, TINY_CONJ.ctx_tokens, dtype=torch.bool, device="cuda")
    y_eq = m(G._preprocess(x[:2]), mask_eq, x_context=imu[:2], mask_context=mc_eq).cpu().numpy()
    assert np.abs(y_eq - g["y_tokens_eq"]).max() <= 3e-4
    for k in ("padding_mask", "full_input_mask", "null_mask"):
        assert np.array_equal(getattr(m.main_stream, k).cpu().numpy(), pad[k + "_eq"]), k
    video = G.predict(x[:2], mask_eq.clone(), frame=None, x_context=imu[:2], mask_context=mc_eq).cpu().numpy()
    assert video.shape == g["video_eq"].shape and np.abs(video - g["video_eq"]).max() <= 3e-4
    # fast mode stays close on this shallow model
    m.mode = "fast"
    yf = m(G._preprocess(x), mask, x_context=imu, mask_context=mc).cpu().numpy()
    assert np.abs(yf - g["y_tokens"]).max() <= 8e-2


def test_imu400_full_size_golden():
    g = np.load(os.path.join(GOLDEN, "conj_imu400_b2.npz"))
    cfg = C.CONJ_CONFIGS["imu400_base_4x4patch_2frames_1tube"]
    m = build(cfg, int(g["seed"]))
    x = torch.from_numpy(S.synthetic_frames(2, cfg.main, 0))
    mask, imu = torch.from_numpy(g["mask"]).cuda(), torch.from_numpy(g["imu"]).cuda()
    y = m(V.preprocess(x).cuda(), mask, x_context=imu, mask_context=torch.zeros(2, 25, dtype=torch.bool, device="cuda")).cpu().numpy()
    err = np.abs(y - g["y_tokens"]).max()
    print(f"[imu400 B=2 ragged] parity-mode max-abs vs reference: {err:.3e}")
    assert y.shape == g["y_tokens"].shape and err <= 1e-3, err
    assert np.array_equal(np.abs(y).sum(-1) == 0, np.abs(g["y_tokens"]).sum(-1) == 0)
    m.mode = "fast"
    yf = m(V.preprocess(x).cuda(), mask, x_context=imu, mask_context=torch.zeros(2, 25, dtype=torch.bool, device="cuda")).cpu().numpy()
    print(f"[imu400 B=2 ragged] fast-mode max-abs vs reference: {np.abs(yf - g['y_tokens']).max():.3e}")
    assert np.abs(yf - g["y_tokens"]).max() <= 2.5e-1


def test_imu400_sharp_weights_golden():
    """The IMU-conditioned model under the hostile weights (sharp softmax in both streams' self-attention, LayerNorm weights U(0.2, 3)
    incl. the cross blocks' norms, residual growth), one masked context token, against the reference's output."""
    g = np.load(os.path.join(GOLDEN, "conj_imu400_sharp_b1.npz"))
    cfg = C.CONJ_CONFIGS["imu400_base_4x4patch_2frames_1tube"]
    seed = int(g["seed"])
    m = build(cfg, seed, sharp=True)
    x = torch.from_numpy(S.synthetic_frames(1, cfg.main, seed))
    mask, imu, mc = (torch.from_numpy(g[k]).cuda() for k in ("mask", "imu", "mask_context"))
    y = m(V.preprocess(x).cuda(), mask, x_context=imu, mask_context=mc).cpu().numpy()
    err = np.abs(y - g["y_tokens"]).max()
    print(f"[imu400 sharp] parity-mode max-abs vs reference: {err:.3e}")
    assert y.shape == g["y_tokens"].shape and err <= 1e-3, err
    assert np.array_equal(np.abs(y).sum(-1) == 0, np.abs(g["y_tokens"]).sum(-1) == 0)


def test_context_stream_output_goldens():
    """`forward(..., output_context=True)` (conjoined_vmae.py:852-887, 990-1011): the context stream's predictions over its masked + pad
    slots, alone or in a tuple with the main output, against the reference (tiny model: ragged visible counts, masked IMU tokens; full
    size: three masked IMU tokens); the flags are sticky like the reference's `_set_decoder_outputs` (:589-593)."""
    g = np.load(os.path.join(GOLDEN, "conj_tiny_ctx.npz"))
    m = build(TINY_CONJ, int(g["seed"]))
    G = prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
    x, mask, imu, mc = (torch.from_numpy(g[k]).cuda() for k in ("x", "mask", "imu", "mask_context"))
    xp = G._preprocess(x)
    y0 = m(xp, mask, x_context=imu, mask_context=mc)                       # default: main output only
    assert torch.is_tensor(y0)
    out = m(xp, mask, x_context=imu, mask_context=mc, output_main=True, output_context=True)
    assert isinstance(out, tuple) and len(out) == 2
    y, y_c = (t.cpu().numpy() for t in out)
    assert torch.equal(out[0], y0)
    assert y.shape == g["y_tokens"].shape and np.abs(y - g["y_tokens"]).max() <= 3e-4
    assert y_c.shape == g["y_ctx_tokens"].shape and np.abs(y_c - g["y_ctx_tokens"]).max() <= 3e-4
    assert np.array_equal(np.abs(y_c).sum(-1) == 0, g["ctx_null_mask"])     # zero rows = masked pad slots of the context stream
    y_c2 = m(xp, mask, x_context=imu, mask_context=mc, output_main=False)    # context alone ...
    assert torch.is_tensor(y_c2) and torch.equal(y_c2, out[1])
    y_c3 = m(xp, mask, x_context=imu, mask_context=mc)                      # ... and the setting sticks
    assert torch.is_tensor(y_c3) and torch.equal(y_c3, out[1])
    both = m(xp, mask, x_context=imu, mask_context=mc, output_main=False, output_context=False)   # "all the tokens from both streams"
    assert isinstance(both, tuple) and torch.equal(both[0], out[0]) and torch.equal(both[1], out[1])
    m(xp, mask, x_context=imu, mask_context=mc, output_main=True)          # back to the default for whoever uses the model next
    # full size
    g = np.load(os.path.join(GOLDEN, "conj_imu400_ctx_b1.npz"))
    cfg = C.CONJ_CONFIGS["imu400_base_4x4patch_2frames_1tube"]
    m = build(cfg, int(g["seed"]))
    x = torch.from_numpy(S.synthetic_frames(1, cfg.main, int(g["frames_seed"])))
    mask, imu, mc = (torch.from_numpy(g[k]).cuda() for k in ("mask", "imu", "mask_context"))
    y, y_c = m(V.preprocess(x).cuda(), mask, x_context=imu, mask_context=mc, output_main=True, output_context=True)
    err = np.abs(y_c.cpu().numpy() - g["y_ctx_tokens"]).max()
    print(f"[conj ctx output, full size] max-abs vs reference {err:.2e}")
    assert y_c.shape == (1, 28, 96) and err <= 1e-3
    assert np.abs(y[:, :8].cpu().numpy() - g["y_tokens_head"]).max() <= 1e-3
    yd = y.double()
    assert abs(yd.abs().sum().item() - g["y_tokens_digest"][1]) <= 1e-4 * g["y_tokens_digest"][1]   # (the plain sum cancels: absolute bound)
    assert abs(yd.sum().item() - g["y_tokens_digest"][0]) <= 1e-4 * yd.numel()
    assert (y_c[0, 3:] == 0).all()


def test_conj_errors():
    m = build(TINY_CONJ, 5)
    x = torch.zeros(2, 3, 2, 32, 32, device="cuda")
    mask = torch.zeros(2, 128, dtype=torch.bool, device="cuda")
    mask[:, 64:] = True
    imu = torch.zeros(2, 6, 64, device="cuda")
    with pytest.raises(RuntimeError):
        m(x, mask)  # no IMU
    bad = mask.clone()
    bad[0, 64:80] = False  # 16 more visible than row 1: exceeds max_padding_tokens = 8
    with pytest.raises(RuntimeError):
        m(x, bad, x_context=imu)
    with pytest.raises(RuntimeError):
        m(x.cpu(), mask.cpu(), x_context=imu.cpu())


def test_imu400_two_lanes_match_one_lane():
    """cwm_conj_set_lanes: batch 5 of the full-size IMU model (ragged visible counts, ragged context masks) runs as 3 + 2 on two
    streams; both halves keep the call's n_vis_max, so every row -- values and the zeroed pad slots -- must equal the one-lane result."""
    g = np.load(os.path.join(GOLDEN, "conj_imu400_b2.npz"))
    cfg = C.CONJ_CONFIGS["imu400_base_4x4patch_2frames_1tube"]
    m = build(cfg, int(g["seed"]))
    B = 5
    x = V.preprocess(torch.from_numpy(S.synthetic_frames(B, cfg.main, 1))).cuda()
    mask2 = torch.from_numpy(g["mask"])                       # two rows with different visible counts
    mask = torch.stack([mask2[i % 2] for i in range(B)]).cuda()
    imu = torch.from_numpy(g["imu"])
    imu = torch.stack([imu[i % 2] * (1.0 + 0.1 * i) for i in range(B)]).cuda()
    mc = torch.zeros(B, 25, dtype=torch.bool, device="cuda")
    mc[1, 3] = True
    mc[4, 7:9] = True
    y2 = m(x, mask, x_context=imu, mask_context=mc)           # library default: two lanes (3 + 2)
    m.set_lanes(1)
    y1 = m(x, mask, x_context=imu, mask_context=mc)
    m.set_lanes(2)
    err = (y2 - y1).abs().max().item()
    print(f"[imu400 lanes] 2 vs 1 lanes: {err:.2e}")
    assert y1.shape == y2.shape and err <= 5e-5
    assert torch.equal(y1.abs().sum(-1) == 0, y2.abs().sum(-1) == 0)
    assert torch.equal(m(x, mask, x_context=imu, mask_context=mc), y2)


@pytest.mark.parametrize("which", ["tiny", "imu400"])
def test_mfma_and_valu_conj_attention_agree(which):
    """The cross attention / context self-attention on MFMAs (csrc/conj_attention.hip: flash dataflow, split-bf16 products, the
    projections in the GEMM operand layout) against the exact-fp32 VALU kernels they replace (csrc/conj_kernels.hip, debug switch
    "conj_attn" = 0): same forward output within the split-bf16 rounding, for ragged visible counts, masked context tokens
    (fewer than 25 context keys), head_dim 32 (tiny) and 192 / 96 with 25 / 50 context tokens (imu400), in both modes."""
    if which == "tiny":
        g = np.load(os.path.join(GOLDEN, "conj_tiny.npz"))
        m = build(TINY_CONJ, int(g["seed"]))
        x = V.preprocess(torch.from_numpy(g["x"])).cuda()
        mask, imu, mc = (torch.from_numpy(g[k]).cuda() for k in ("mask", "imu", "mask_context"))
    else:
        g = np.load(os.path.join(GOLDEN, "conj_imu400_b2.npz"))
        cfg = C.CONJ_CONFIGS["imu400_base_4x4patch_2frames_1tube"]
        m = build(cfg, int(g["seed"]))
        x = V.preprocess(torch.from_numpy(S.synthetic_frames(3, cfg.main, 1))).cuda()
        mask2 = torch.from_numpy(g["mask"])
        mask = torch.stack([mask2[i % 2] for i in range(3)]).cuda()
        imu = torch.stack([torch.from_numpy(g["imu"])[i % 2] * (1.0 + 0.3 * i) for i in range(3)]).cuda()
        mc = torch.zeros(3, 25, dtype=torch.bool, device="cuda")
        mc[1, 3] = True
        mc[2, 7:12] = True
    for mode, tol in (("parity", 1e-4), ("fast", 1e-1)):
        m.mode = mode
        m.set_option("conj_attn", 1)
        y1 = m(x, mask, x_context=imu, mask_context=mc)
        m.set_option("conj_attn", 0)
        y0 = m(x, mask, x_context=imu, mask_context=mc)
        err = (y1 - y0).abs().max().item()
        print(f"[{which} {mode}] MFMA vs VALU conj attention: {err:.2e}")
        assert torch.isfinite(y1).all() and err <= tol, (mode, err)
        assert torch.equal(y1.abs().sum(-1) == 0, y0.abs().sum(-1) == 0)


def test_context_side_stream_is_bitwise_neutral():
    """Option "conj_ctx_stream": the IMU stream's blocks and the context side of every cross block run on a side HIP stream per
    lane, exchanging projections with the lane's stream through events (csrc/conj_model.hip run_cross).  Pure scheduling: outputs must be
    bit-identical to the one-stream order, call after call (a missing event would show as a race), for one and two lanes."""
    g = np.load(os.path.join(GOLDEN, "conj_imu400_b2.npz"))
    cfg = C.CONJ_CONFIGS["imu400_base_4x4patch_2frames_1tube"]
    m = build(cfg, int(g["seed"]))
    B = 6
    x = V.preprocess(torch.from_numpy(S.synthetic_frames(B, cfg.main, 2))).cuda()
    mask = torch.stack([torch.from_numpy(g["mask"])[i % 2] for i in range(B)]).cuda()
    imu = torch.stack([torch.from_numpy(g["imu"])[i % 2] * (1.0 + 0.2 * i) for i in range(B)]).cuda()
    mc = torch.zeros(B, 25, dtype=torch.bool, device="cuda")
    mc[2, 5] = True
    m(x[:1], mask[:1], x_context=imu[:1], mask_context=mc[:1])   # creates the library handle
    for lanes in (2, 1):
        m.set_lanes(lanes)
        m.set_option("conj_ctx_stream", 0)
        ref = m(x, mask, x_context=imu, mask_context=mc)
        m.set_option("conj_ctx_stream", 1)
        for rep in range(4):
            assert torch.equal(m(x, mask, x_context=imu, mask_context=mc), ref), (lanes, rep)
    for lanes in (0, 3, 4):   # the conjoined model runs one or two lanes (each lane carries a context stream: four queues); the setter accepts nothing else
        with pytest.raises(RuntimeError, match="lanes must be 1 or 2"):
            m.set_lanes(lanes)


def test_forward_args_struct_size_versions():
    """`struct_size` (ABI 0.6): a caller compiled against the 0.4 header -- whose `cwm_conj_forward_args` ended at `stream`, before `y_ctx_tokens_dev` was appended -- passes the
    shorter struct and gets the main output (the library copies what the caller has and reads the rest as "not requested", instead of taking stack garbage for an output pointer);
    the full struct also fills the context output; sizes that are no version of the struct are refused.  Same for `cwm_forward_args`."""
    import ctypes as Ct

    from counterfactualworldmodels_amd import _lib

    g = np.load(os.path.join(GOLDEN, "conj_tiny_ctx.npz"))
    m = build(TINY_CONJ, int(g["seed"]))
    G = prediction.PredictorBasedGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
    x, mask, imu, mc = (torch.from_numpy(g[k]).cuda() for k in ("x", "mask", "imu", "mask_context"))
    xp = G._preprocess(x).contiguous()
    y_ref, yc_ref = m(xp, mask, x_context=imu, mask_context=mc, output_main=True, output_context=True)
    m(xp, mask, x_context=imu, mask_context=mc, output_main=True, output_context=False)
    lib, h = _lib.get_lib(), m._handle
    vis, vis_c = (~mask).sum(1), (~mc).sum(1)
    y, yc = torch.full_like(y_ref, 7.0), torch.full_like(yc_ref, 7.0)
    full = _lib.CwmConjForwardArgs(Ct.sizeof(_lib.CwmConjForwardArgs), xp.data_ptr(), xp.stride(0), xp.stride(1), xp.stride(2), 0, mask.data_ptr(), xp.shape[0], int(vis.max()),
                                   imu.data_ptr(), mc.data_ptr(), int(vis_c.max()), y.data_ptr(), _lib.MODE_PARITY, 1, _lib.current_stream_handle(xp.device), yc.data_ptr())
    assert lib.cwm_conj_forward(h, Ct.byref(full)) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref) and torch.equal(yc, yc_ref)
    # the 0.4 layout: everything up to and including `stream`; the bytes behind it in the caller's frame are garbage the library must not read
    old_size = _lib.CwmConjForwardArgs.stream.offset + Ct.sizeof(Ct.c_void_p)
    y.fill_(7.0)
    yc.fill_(7.0)
    full.struct_size = old_size
    full.y_ctx_tokens_dev = 0xDEAD0000  # (what a short struct's tail would look like to a library that ignored the size)
    assert lib.cwm_conj_forward(h, Ct.byref(full)) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref) and bool((yc == 7.0).all())
    for bad in (0, 8, old_size - 8, 1 << 20):
        full.struct_size = bad
        assert lib.cwm_conj_forward(h, Ct.byref(full)) == _lib.ERR_INVALID and b"struct_size" in lib.cwm_last_error()
    # the plain model's struct
    from test_model_gpu import TINY, build as build_plain, case_inputs

    gp = np.load(os.path.join(GOLDEN, "tiny_8x8_k4.npz"))
    seed, xs, ms = case_inputs(gp, TINY)
    mp = build_plain(TINY, seed)
    Gp = prediction.PredictorBasedGenerator(predictor=mp, imagenet_normalize_inputs=True, temporal_dim=2)
    xq, mq = Gp._preprocess(xs.cuda()).contiguous(), ms.cuda().contiguous()
    yr = mp(xq, mq)
    yo = torch.empty_like(yr)
    a = _lib.CwmForwardArgs(Ct.sizeof(_lib.CwmForwardArgs), xq.data_ptr(), xq.stride(0), xq.stride(1), xq.stride(2), 0, mq.data_ptr(), xq.shape[0], int((~mq[0]).sum()),
                            yo.data_ptr(), None, None, _lib.MODE_PARITY, 1, _lib.current_stream_handle(xq.device))
    assert lib.cwm_forward(mp._handle, Ct.byref(a)) == 0
    torch.cuda.synchronize()
    assert torch.equal(yo, yr)
    a.struct_size = 0
    assert lib.cwm_forward(mp._handle, Ct.byref(a)) == _lib.ERR_INVALID
