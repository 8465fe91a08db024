"""f-1 / f-2: device-side motion-counterfactual prompt construction against the fixture captured
from the reference's own `FlowGenerator.create_motion_counterfactuals` (bit-exact), and the batched
counterfactual prediction driver."""
import os

import numpy as np
import pytest
import torch

from counterfactualworldmodels_amd import config as C, conjoined_vmae as CV, dist as cdist, segmentation, synthetic as S, vmae
from oracle import vmae_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TINY = C.VmaeConfig(name="tiny_8x8", img_size=(32, 32), patch=8, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128,
                    dec_depth=1, dec_heads=2)
TINY_SPEC = O.VmaeSpec(img_size=(32, 32), patch=8, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128, dec_depth=1, dec_heads=2)


def wrapper(cfg, seed=3, **kw):
    m = vmae.PretrainVisionTransformer(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, seed).items()})
    return segmentation.FlowGenerator(predictor=m.cuda().eval(), imagenet_normalize_inputs=True, temporal_dim=2, **kw)


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


class DummyFlow(torch.nn.Module):
    """The stand-in flow model of tests/golden/make_golden.py `run_wrapper_cases` (the reference plugs RAFT in here)."""

    def forward(self, x, backward=False, **k):
        d = x[:, 1:] - x[:, :-1]
        return torch.stack([d.mean(2), d.amax(2)], 2)


def test_single_prompt_counterfactual_and_error_maps_vs_reference():
    """`get_counterfactual_prediction` (the UI's click, interface.py:273-299), `_shift`, `predict_error` against outputs of the
    reference wrapper (wrapper_surface.npz): prompt frames / masks bit-exact, predictions within tolerance."""
    g = np.load(os.path.join(GOLDEN, "wrapper_surface.npz"))
    G = wrapper(TINY, 3)
    img, passive, active = (torch.from_numpy(g[k]).cuda() for k in ("cf_img", "cf_passive", "cf_active"))
    xs = G.make_static_movie(img[:, None], T=2)
    x_p, mask_p = G._shift(xs, passive.clone(), active_patches=active.clone(), shift=(1, -1), frame=1)
    assert np.array_equal(x_p.cpu().numpy(), g["cf_x_p"]) and np.array_equal(mask_p.cpu().numpy(), g["cf_mask_p"])
    G.shifts = None
    y = G.get_counterfactual_prediction(img, mask=passive.clone(), active_patches=active.clone(), shift=(1, -1))
    assert y.shape == g["cf_y"].shape and np.abs(y.cpu().numpy() - g["cf_y"]).max() <= 2e-4
    assert np.array_equal(np.array(G.shifts), g["cf_shifts"])
    x = torch.from_numpy(S.synthetic_frames(2, TINY, 41)).cuda()
    mask = torch.from_numpy(g["err_mask"]).cuda()
    e1 = G.predict_error(x, mask.clone(), frame=1)
    ea = G.predict_error(x, mask.clone(), frame=None)
    assert e1.shape == g["err_frame1"].shape and np.abs(e1.cpu().numpy() - g["err_frame1"]).max() <= 2e-4
    assert ea.shape == g["err_all"].shape and np.abs(ea.cpu().numpy() - g["err_all"]).max() <= 2e-4


def test_make_static_on_passive_patches_vs_reference():
    """`get_counterfactual_prediction(fix_passive=True)` = `MakeStatic` on the patches the passive mask leaves visible, then the shift
    (prediction.py:802-812, perturbation.py:120-145), folded into the prompt kernel (fix_passive = 2): frames and masks bit-exact
    against the reference's outputs on a movie whose two frames differ, the prediction within tolerance."""
    g = np.load(os.path.join(GOLDEN, "wrapper_surface.npz"))
    G = wrapper(TINY, 3)
    movie, passive, active = (torch.from_numpy(g[k]).cuda() for k in ("ms_movie", "ms_passive", "cf_active"))
    xs, m = G.make_static(movie, passive.clone())
    assert np.array_equal(xs.cpu().numpy(), g["ms_x"]) and np.array_equal(m.cpu().numpy(), g["ms_mask"])
    x_p, mask_p = G._shift(movie, passive.clone(), active_patches=active.clone(), shift=(1, -1), frame=1, fix_passive="make_static")
    assert np.array_equal(x_p.cpu().numpy(), g["ms_x_p"]) and np.array_equal(mask_p.cpu().numpy(), g["ms_mask_p"])
    G.shifts = None
    y = G.get_counterfactual_prediction(movie, mask=passive.clone(), active_patches=active.clone(), shift=(1, -1), fix_passive=True)
    assert y.shape == g["ms_y"].shape and np.abs(y.cpu().numpy() - g["ms_y"]).max() <= 2e-4
    # and the oracle's restatement on a random table at the B/8 grid: the two-step form equals the fused kernel
    cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
    Gb = wrapper(cfg, 0)
    xb = torch.from_numpy(S.synthetic_frames(2, cfg, 9)).cuda()
    gen = torch.Generator().manual_seed(3)
    pas = torch.rand(2, cfg.num_tokens, generator=gen) < 0.9
    pas[:, : cfg.tokens_per_frame] = False
    act = torch.ones(2, cfg.num_tokens, dtype=torch.bool)
    act[0, cfg.tokens_per_frame + 100] = False
    act[1, cfg.tokens_per_frame + 333] = False
    xs_ref = O.make_static(xb.cpu(), pas, cfg.patch)
    xs_dev, _ = Gb.make_static(xb, pas.cuda())
    assert torch.equal(xs_dev.cpu(), xs_ref)
    for b in range(2):
        xr, mr = O.shift_patches_and_mask(xs_ref[b:b + 1], pas[b:b + 1], act[b:b + 1], (2, -3), cfg.patch, frame=1)
        xd, md = Gb._shift_rows(xb[b:b + 1], pas[b:b + 1].cuda(), act[b:b + 1].cuda(), torch.tensor([[2, -3]], dtype=torch.int32), 1, "make_static")
        assert torch.equal(xd.cpu(), xr) and torch.equal(md.cpu(), mr)


def test_counterfactual_videos_and_flows_vs_reference():
    """`predict_counterfactual_videos_and_flows` with a plugged flow model (segmentation.py:346-432): '(b s)' ordering, the shifts
    list, flows [B*S,1,2,H,W], values equal to the reference's own run of the same call; independent of sample_batch_size."""
    g = np.load(os.path.join(GOLDEN, "wrapper_surface.npz"))
    G = wrapper(TINY, 3, flow_model=DummyFlow())
    img = torch.from_numpy(g["cf_img"]).cuda()
    act = torch.from_numpy(g["drv_active"]).cuda()
    shifts = [list(int(v) for v in r) for r in g["drv_shifts"]]
    outs = {}
    for sbs in (64, 2, None):
        torch.manual_seed(5)
        ys, fs = G.predict_counterfactual_videos_and_flows(img, active_patches=act.clone(), shifts=shifts, num_samples=5, sample_batch_size=sbs)
        outs[sbs] = (ys, fs)
        assert ys.shape == g["drv_ys"].shape and fs.shape == g["drv_flows"].shape == (5, 1, 2, 32, 32)
        assert np.abs(ys.cpu().numpy() - g["drv_ys"]).max() <= 2e-4 and np.abs(fs.cpu().numpy() - g["drv_flows"]).max() <= 4e-4
        assert np.array_equal(np.array(G.shifts), g["drv_shift_list"])
    assert (outs[64][0] - outs[2][0]).abs().max().item() <= 1e-5 and (outs[64][0] - outs[None][0]).abs().max().item() <= 1e-5
    with pytest.raises(RuntimeError, match="flow_model"):
        wrapper(TINY, 3).predict_counterfactual_videos_and_flows(img, active_patches=act.clone(), shifts=shifts, num_samples=5)
    # tensor form of the shifts argument ([2, S]) and per-sample flow layout helper
    torch.manual_seed(5)
    ys_t = G.predict_counterfactual_videos(img, act.clone(), shifts=torch.tensor(g["drv_shifts"]).T, num_samples=5, sample_batch_size=3)
    assert torch.equal(ys_t, outs[64][0]) or (ys_t - outs[64][0]).abs().max().item() <= 1e-5
    assert G._batch_to_samples(outs[64][1]).shape == (1, 2, 32, 32, 5)


def test_imu_conditioned_driver_vs_reference():
    """The IMU override (segmentation.py:931-963): head motion forwarded as x_context / mask_context through the batch driver, tiled over
    the S prompts of its movie; against the reference's run on the tiny conjoined model (wrapper_conj.npz) for several chunkings."""
    from test_conj_oracle import TINY_CONJ, conj_weights

    g = np.load(os.path.join(GOLDEN, "wrapper_conj.npz"))
    m = CV.ConjoinedPaddedVisionTransformer(TINY_CONJ)
    m.load_state_dict(conj_weights(TINY_CONJ, int(g["seed"])))
    G = segmentation.ImuConditionedFlowGenerator(predictor=m.cuda().eval(), flow_model=DummyFlow(), imagenet_normalize_inputs=True, temporal_dim=2)
    assert G.num_head_tokens == TINY_CONJ.ctx_tokens and G.head_motion_channels == 6
    img, imu, act = (torch.from_numpy(g[k]).cuda() for k in ("img", "imu", "active"))
    shifts = [list(int(v) for v in r) for r in g["shifts"]]
    for sbs in (64, 3, 1):
        torch.manual_seed(6)
        ys, fs = G.predict_counterfactual_videos_and_flows(img, active_patches=act.clone(), shifts=shifts, num_samples=4, sample_batch_size=sbs,
                                                           head_motion=imu, mask_head_motion=False, static_head_motion=True)
        assert ys.shape == g["ys"].shape and fs.shape == g["flows"].shape
        err = np.abs(ys.cpu().numpy() - g["ys"]).max()
        assert err <= 3e-4, (sbs, err)
        assert not hasattr(m, "padding_mask")   # padding state reset after the call (prediction.py:451-452)
    with pytest.raises(RuntimeError, match="head_motion"):
        G.predict_counterfactual_videos(img, act.clone(), shifts=shifts, num_samples=4)
    # two movies: each movie's prompts get that movie's IMU
    img2 = torch.cat([img, img.flip(-1)], 0)
    imu2 = torch.cat([imu, imu * -1.0], 0)
    act2 = act.expand(2, -1, -1)
    torch.manual_seed(6)
    y2 = G.predict_counterfactual_videos(img2, act2.clone(), shifts=shifts, num_samples=4, sample_batch_size=8, head_motion=imu2)
    assert y2.shape == (8, 2, 3, 32, 32) and np.abs(y2[:4].cpu().numpy() - g["ys"]).max() <= 3e-4
    torch.manual_seed(6)
    y2b = G.predict_counterfactual_videos(img2[1:], act2[1:].clone(), shifts=shifts, num_samples=4, sample_batch_size=8, head_motion=imu2[1:])
    assert (y2[4:] - y2b).abs().max().item() <= 1e-5


def test_prompt_table_expand_and_one_frame_movie_are_bit_exact():
    """`cwm_prompt_table_expand` (prompt table -> dense active / passive masks + shifts in one launch) against the tensor operations it replaced, and the
    `num_frames` form of the prompt kernel (a static movie given as frame 0 alone) against the materialised two-frame movie: same frames, same masks, bit for bit;
    masks-only / frames-only calls give the same tensors as the combined call."""
    import ctypes as C_

    from counterfactualworldmodels_amd import _lib

    cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
    G = wrapper(cfg, 0)
    S_ = 37
    table = torch.from_numpy(S.synthetic_prompts(S_, cfg, 5)).cuda()
    table[3, 2:4] = torch.tensor([-30, 2])      # a shift that leaves the frame
    gh = gw = cfg.img_size[0] // cfg.patch
    n = gh * gw
    active = torch.empty(S_, 2 * n, dtype=torch.bool, device="cuda")
    passive = torch.empty_like(active)
    shifts = torch.empty(S_, 2, dtype=torch.int32, device="cuda")
    _lib.check(_lib.get_lib().cwm_prompt_table_expand(table.data_ptr(), S_, 2, gh, gw, 1, active.data_ptr(), passive.data_ptr(), shifts.data_ptr(),
                                                     _lib.current_stream_handle(torch.device("cuda:0"))))
    ref_passive = (torch.arange(2 * n, device="cuda") >= n)[None].expand(S_, -1)
    ref_active = ref_passive.clone()
    ref_active[torch.arange(S_, device="cuda"), n + table[:, 0].long() * gw + table[:, 1].long()] = False
    assert torch.equal(passive, ref_passive) and torch.equal(active, ref_active) and torch.equal(shifts, table[:, 2:4])
    x0 = torch.from_numpy(S.synthetic_frames(1, cfg, 0))[:, 0:1].cuda()
    G.inp_shape = (1, 2) + tuple(x0.shape[2:])
    xs_ref, ms_ref = G._shift_rows(x0.expand(-1, 2, -1, -1, -1), ref_passive, ref_active, table[:, 2:4], 1, True, samples_per_movie=S_)
    build, _, _ = cdist.prompt_hooks(G, frame=-1)
    xs, ms = build(x0, table)
    assert torch.equal(xs, xs_ref) and torch.equal(ms, ms_ref)
    none_x, ms_only = build(x0, table, frames=False)
    assert none_x is None and torch.equal(ms_only, ms_ref)
    xs_only, none_m = G._shift_rows(x0, passive, active, shifts, 1, True, samples_per_movie=S_, masks=False, num_frames=2)
    assert none_m is None and torch.equal(xs_only, xs_ref)
    with pytest.raises(RuntimeError):
        G._shift_rows(x0, passive, active, shifts, 1, False, samples_per_movie=S_, num_frames=2)   # a one-frame movie is only a static movie


def test_256_prompts_sharded_driver_equals_one_unchunked_call():
    """BASELINE configs[3] at full size on one rank: 256 prompts on one frame pair through `dist.sharded_counterfactual_predictions`
    (8 library calls of 32 rows, no host sync between them) = ONE 256-row predictor call; plus rows vs the wrapper's own driver."""
    cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
    G = wrapper(cfg, 0)
    x0 = torch.from_numpy(S.synthetic_frames(1, cfg, 0))[:, 0:1]
    table = torch.from_numpy(S.synthetic_prompts(256, cfg, 0))
    hooks = cdist.prompt_hooks(G, frame=-1)
    torch.manual_seed(3)
    y = cdist.sharded_counterfactual_predictions(x0, table, *hooks, torch.device("cuda:0"), chunk=32, comm=cdist.LocalComm())
    assert y.shape == (256, 1, 3, 224, 224) and torch.isfinite(y).all()
    torch.manual_seed(3)
    y_one = cdist.sharded_counterfactual_predictions(x0, table, *hooks, torch.device("cuda:0"), chunk=256, comm=cdist.LocalComm())
    err = (y - y_one).abs().max().item()
    print(f"[prompts256] 8 x 32 rows vs one 256-row call: {err:.2e}")
    assert err <= 5e-5
    # anchored on the reference: its own prompt construction over these 256 prompts (one rectangularisation under the same torch seed)
    # and its `predict` on three of them -- the first, the first whose shift leaves the frame, the last (tests/golden/make_golden.py --only r4)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prompts256_rows.npz"))
    rows = [int(r) for r in g["rows"]]
    assert int(g["seed"]) == 3 and rows[0] == 0 and rows[-1] == 255
    err_ref = np.abs(y[rows][:, :, :, ::2].cpu().numpy() - g["y_rows_even"]).max()
    print(f"[prompts256] rows {rows} vs the reference's predict: {err_ref:.2e}")
    assert err_ref <= 1e-3
    # the same prompts through the reference-shaped entry point (active patch masks + shifts)
    n = cfg.tokens_per_frame
    gw = cfg.img_size[1] // cfg.patch
    sub = slice(0, 16)
    active = torch.ones(1, 2 * n, 16, dtype=torch.bool)
    active[:, :n] = False
    cell = n + table[sub, 0].long() * gw + table[sub, 1].long()
    active[0, cell, torch.arange(16)] = False
    torch.manual_seed(3)
    yv = G.predict_counterfactual_videos(x0[:, 0].cuda(), active.cuda(), shifts=table[sub, 2:4].tolist(), num_samples=16, sample_batch_size=32)
    assert (yv[:, 1:] - y[sub]).abs().max().item() <= 5e-5
    assert torch.equal(yv[:, 0], x0[0, 0].cuda().expand(16, -1, -1, -1))   # frame 0 is the input image


def test_rectangulariser_on_device_masks_is_bit_exact_and_in_place():
    """`RectangularizeMasks` on device masks (pinned host staging, numpy row edits, one copy back): same result as on the CPU tensor with the
    same torch seed (the reference's randperm order, masking.py:100-132), input mutated in place, `last_num_masked` reported.
    NB this is a SELF-comparison (device-staged path vs the same class on a CPU tensor); the pin against the reference for this row
    is the CPU test against the reference's own output, tests/test_host_logic.py (index_ops.npz `rect_out`)."""
    from counterfactualworldmodels_amd.masking import RectangularizeMasks

    g = torch.Generator().manual_seed(4)
    for mode in ("min", "max", "mean"):
        for rows in (1, 5, 64):
            m = torch.rand(rows, 200, generator=g) < 0.7
            torch.manual_seed(77)
            ref = RectangularizeMasks(mode)(m.clone())
            torch.manual_seed(77)
            md = m.cuda()
            r = RectangularizeMasks(mode)
            out = r(md)
            assert out.data_ptr() == md.data_ptr() and torch.equal(out.cpu(), ref)
            assert r.last_num_masked == int(ref[0].sum()) and (ref.sum(-1) == r.last_num_masked).all()
            # the read-back polls an event by default (masking.py _to_host); the blocking copy gives the same masks, also behind queued work
            torch.manual_seed(77)
            mb = m.cuda()
            busy = torch.randn(2048, 2048, device="cuda")
            for _ in range(4):
                busy = busy @ busy * 1e-3
            rb = RectangularizeMasks(mode)
            rb.spin_wait = False
            assert torch.equal(rb(mb).cpu(), ref) and rb.last_num_masked == r.last_num_masked


def test_rccl_comm_single_rank_on_device():
    """The C-ABI collectives (cwm_comm_init / cwm_broadcast / cwm_allgatherv / cwm_allreduce_sum_f32) on a one-rank RCCL communicator:
    RCCL binds, the communicator initialises on the GPU, and every collective is the identity on the stream."""
    from counterfactualworldmodels_amd import _lib

    assert _lib.get_lib().cwm_comm_version() >= 22000
    comm = cdist.RcclComm(0, 1, cdist.RcclComm.new_unique_id(), torch.device("cuda:0"))
    try:
        buf = torch.arange(4096, dtype=torch.uint8, device="cuda")
        ref = buf.clone()
        comm.broadcast_bytes(buf, 0)
        local = torch.randn(5, 7, 3, device="cuda")
        out = torch.empty_like(local)
        comm.all_gather_blocks(local, out, [5])
        t = torch.randn(1000, device="cuda")
        t0 = t.clone()
        comm.all_reduce_sum(t)
        torch.cuda.synchronize()
        assert torch.equal(buf, ref) and torch.equal(out, local) and torch.equal(t, t0)
        # the sharded driver over this communicator (world 1 short-circuits the collectives but not the plumbing)
        G = wrapper(TINY, 3)
        x0 = torch.from_numpy(S.synthetic_frames(1, TINY, 2))[:, 0:1]
        table = torch.tensor([[1, 2, 1, 0], [3, 3, 0, -1], [0, 0, -1, -1]], dtype=torch.int32)
        y = cdist.sharded_counterfactual_predictions(x0, table, *cdist.prompt_hooks(G), torch.device("cuda:0"), chunk=2, comm=comm)
        assert y.shape == (3, 1, 3, 32, 32)
    finally:
        comm.close()


def test_rccl_gather_block_layouts_on_one_rank():
    """What breaks first on a multi-GPU node is pointer arithmetic.  The 8-rank / 256-prompt gather is `ncclAllGather` in place over 8 equal
    blocks of 32 rows, rank r's block at row 32 r (dist.py `RcclComm.gather_args`); the other shapes take the grouped-broadcast
    `cwm_allgatherv` with per-rank byte offsets.  One GPU cannot host 8 ranks (RCCL refuses two ranks on a device), so: (a) the pure pointer
    function is checked for every rank of the 8-rank layout (tests/test_dist_cpu.py does the same without a GPU), and (b) the REAL collectives
    run on a one-rank communicator whose single block sits where rank r's would -- rows 32 r .. 32 r + 31 of the 256-row result, and ragged
    offsets for the allgatherv form -- and must leave that block intact and every other row untouched (an off-by-a-block send / receive
    pointer would move data)."""
    comm = cdist.RcclComm(0, 1, cdist.RcclComm.new_unique_id(), torch.device("cuda:0"))
    try:
        rows, width = 256, 3 * 224 * 2  # (a narrow stand-in for the [256, 1, 3, 224, 224] result: the arithmetic is in rows)
        for r in range(8):
            out = torch.full((rows, width), -1.0, device="cuda")
            block = torch.randn(32, width, device="cuda")
            out[32 * r:32 * r + 32] = block
            ref = out.clone()
            kind, send, recv, nbytes = cdist.RcclComm.gather_args(out.data_ptr(), width * 4, [32 * q for q in range(8)], [32] * 8, r)
            assert kind == "allgather" and send == out.data_ptr() + r * 32 * width * 4 and recv == out.data_ptr() and nbytes == 32 * width * 4
            comm.all_gather_rows(out, [32 * r], [32])        # the one-rank communicator's block = rank r's slot
            assert comm.last_collective == "ncclAllGather"
            torch.cuda.synchronize()
            assert torch.equal(out, ref), r
        # ragged: 250 prompts on 8 ranks = 32, 32, 31 ... rows; chunk 1 of a 2-chunk schedule: some ranks have nothing left
        for offs, cnts in (([100], [31]), ([7], [1]), ([255], [1])):
            out = torch.full((rows, width), -1.0, device="cuda")
            out[offs[0]:offs[0] + cnts[0]] = 2.0
            ref = out.clone()
            # (a one-block layout is "equal and contiguous" by definition: force the allgatherv form through the C ABI directly)
            import ctypes as C

            from counterfactualworldmodels_amd import _lib

            sizes = (C.c_size_t * 1)(cnts[0] * width * 4)
            o = (C.c_size_t * 1)(offs[0] * width * 4)
            _lib.check(_lib.get_lib().cwm_allgatherv(comm._handle, out.data_ptr() + offs[0] * width * 4, out.data_ptr(), o, sizes, comm._stream()))
            torch.cuda.synchronize()
            assert torch.equal(out, ref), (offs, cnts)
    finally:
        comm.close()
