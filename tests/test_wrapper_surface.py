"""Wrapper-surface host logic against fixtures captured from the reference classes (tests/golden/make_golden.py
`run_wrapper_cases`): mask generators (bit-exact, torch global RNG + numpy stream), patch-index masks, masked-patch compositing.
CPU only: nothing here launches a kernel."""
import os

import numpy as np
import pytest
import torch

from counterfactualworldmodels_amd import config as C, masking as M, segmentation, vmae

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TINY = C.VmaeConfig(name="tiny_8x8", img_size=(32, 32), patch=8, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128,
                    dec_depth=1, dec_heads=2)


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "wrapper_surface.npz"))


def test_rotated_table_generator_known_answer_and_fixtures(g):
    """masking.py:478-545; the demo notebook's generator: 3104 masked tokens on the (2,56,56) grid with clumps of 2x2."""
    x2 = torch.zeros(2, 2, 3, 8, 8)
    gen = M.RotatedTableUniformMaskingGenerator(input_size=(2, 56, 56), mask_ratio=0.99, clumping_factor=2, randomize_num_visible=False,
                                                always_batch=True, seed=0)
    m0 = gen(None)
    assert m0.shape == (1, 6272) and m0.dtype == torch.bool and int(m0.sum()) == 3104 and not m0[:, :3136].any()
    assert np.array_equal(m0.numpy(), g["gen_rot56_none"])
    assert np.array_equal(gen(x2).numpy(), g["gen_rot56_b2"])
    # visible cells come as 2x2 clumps
    vis = (~m0[0, 3136:]).view(28, 2, 28, 2)
    assert torch.equal(vis.all(3).all(1), vis.any(3).any(1)) and int(vis.all(3).all(1).sum()) == 8
    gen = M.RotatedTableUniformMaskingGenerator(input_size=(3, 7, 7), mask_ratio=0.75, clumping_factor=2, seed=1, randomize_num_visible=True)
    assert np.array_equal(np.stack([gen(x2).numpy() for _ in range(3)]), g["gen_rot7_pad_b2"])     # odd grid: padded row / column
    gen = M.RotatedTableUniformMaskingGenerator(input_size=(2, 8, 8), mask_ratio=0.5, seed=4, full_mask_prob=0.5)
    assert np.array_equal(np.stack([gen(torch.zeros(5, 1)).numpy() for _ in range(2)]), g["gen_rot8_full_b5"])
    gen = M.RotatedTableUniformMaskingGenerator(input_size=(2, 28, 28), mask_ratio=0.9, seed=2)
    gen.num_visible = 3
    out = gen(x2)
    assert np.array_equal(out.numpy(), g["gen_rot28_nv3"]) and int((~out[0, 784:]).sum()) == 3


def test_base_generator_fixture_and_properties(g):
    gen = M.MaskingGenerator(input_size=(1, 7, 7), mask_ratio=0.75, clumping_factor=2, seed=1, visible_frames=1)
    assert np.array_equal(gen(torch.zeros(2, 1)).numpy(), g["gen_base7_b2"])
    gen = M.MaskingGenerator(input_size=(4, 4), mask_ratio=0.5, seed=0)
    assert gen.num_visible == 8 and gen.num_masks_per_frame == 8
    gen.mask_ratio = 0.25
    assert gen.num_masks_per_frame == 4
    m = gen(None)
    assert m.shape == (16,) and int(m.sum()) == 4
    assert gen(torch.zeros(3, 1)).shape == (3, 16)
    assert isinstance(gen, torch.nn.Module)  # the notebook calls .requires_grad_(False).to(device) on it


def _wrapper():
    m = vmae.PretrainVisionTransformer(TINY)
    return segmentation.FlowGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2, seed=0)


def test_generate_mask_from_patch_idx_list(g):
    G = _wrapper()
    G.set_input(torch.zeros(2, 2, 3, 32, 32))
    # The reference writes through the EXPANDED view `get_zeros_mask` returns (deprecated index_put_ on an expanded tensor), so in its
    # output every batch row carries the union of all listed patches; here `b` / the entries' batch index select the row, as the
    # code intends.  The two agree on the addressed rows, and the union over rows is the reference's row.
    hw = G.generate_mask_from_patch_idx_list([[9, 17], [31, 2]], b=1, frame=-1).numpy()
    assert np.array_equal(hw[1], g["idx_hw"][1]) and hw[0, 16:].all()
    thw = G.generate_mask_from_patch_idx_list([[1, 9, 17], [1, 24, 24]], b=0, frame=1).numpy()
    assert np.array_equal(thw[0], g["idx_thw"][0]) and thw[1, 16:].all()
    bthw = G.generate_mask_from_patch_idx_list([[0, 1, 9, 17], [1, 1, 31, 2]], frame=1, stride=8).numpy()
    assert np.array_equal(np.logical_and.reduce(bthw, 0), g["idx_bthw"][0]) and (~bthw[:, 16:]).sum() == 2
    one = _wrapper()
    one.set_input(torch.zeros(1, 2, 3, 32, 32))   # B = 1 (the UI's case): identical to the reference
    assert np.array_equal(one.generate_mask_from_patch_idx_list([[9, 17], [31, 2]], frame=-1).numpy(), g["idx_hw"][:1])
    assert G.generate_mask_from_patch_idx_list([], frame=1).shape == (2, 32)
    z = G.get_zeros_mask()
    assert z.shape == (2, 32) and not z[:, :16].any() and z[:, 16:].all()
    assert not G.get_zeros_mask(frame=None).any()


def test_get_masked_pred_patches(g):
    G = _wrapper()
    G.set_input(torch.zeros(2, 2, 3, 32, 32))
    preds, mask = torch.from_numpy(g["mpp_preds"]), torch.from_numpy(g["mpp_mask"])
    assert np.array_equal(G.get_masked_pred_patches(preds, mask).numpy(), g["mpp_plain"])
    assert np.allclose(G.get_masked_pred_patches(preds, mask, invert=True, fill_value=[0.1, 0.2, 0.3]).numpy(), g["mpp_invert_fill"], atol=1e-7)
    assert np.allclose(G.get_masked_pred_patches(preds, mask, fill_value=1 - preds).numpy(), g["mpp_fill_tensor"], atol=1e-7)
    assert G.inp_shape == (2, 2, 3, 32, 32)  # the call must not disturb the wrapper's input shape


def test_shift_list_forms():
    f = segmentation._shift_list
    assert f([1, -2], 3) == [(1, -2)] * 3
    assert f([[1, 0], [0, 1]], 2) == [(1, 0), (0, 1)]
    assert f(torch.tensor([[1, 2, 3], [4, 5, 6]]), 3) == [(1, 4), (2, 5), (3, 6)]
    assert f(np.array([[1], [4]]), 4) == [(1, 4)] * 4
    with pytest.raises(AssertionError):
        f([[1, 0], [0, 1]], 3)
