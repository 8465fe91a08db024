"""Kernel-level parity on a real MI355X, through the C ABI, against the CPU oracle / torch fp32."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from counterfactualworldmodels_amd import _lib
from oracle import vmae_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# tolerances (max-abs, on O(1) outputs): split-bf16 "parity" mode carries ~2^-16 relative operand
# error; plain bf16 "fast" mode 2^-9 per operand.
TOL = {"parity": 2e-4, "fast": 6e-2}


@pytest.fixture(scope="module")
def gu():
    import gpu_utils

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.get_lib()
    return gpu_utils


@pytest.fixture(scope="module")
def gud():
    """The same entry points bound to the DEVELOPMENT library (libcwm_hip_dev.so = the production objects + csrc/dev.hip): only it exports
    `cwm_debug_set`, which the bitwise cross-checks of kernel variants below need; the switches act on this thread's options inside that shared
    object and never reach the production library."""
    import gpu_utils

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return gpu_utils.Ops(_lib.get_dev_lib())


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("mode", ["parity", "fast"])
@pytest.mark.parametrize(
    "M,N,K",
    [(128, 128, 64), (256, 384, 768), (200, 192, 384), (77, 48, 512), (1000, 1152, 384), (300, 768, 192), (130, 1024, 48), (5, 16, 64)],
)
def test_linear_matches_torch(gu, mode, M, N, K):
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ref = F.linear(a, w, b)
    out = gu.linear(a, w, b, mode=mode)
    err = (out - ref).abs().max().item()
    assert err <= TOL[mode] * max(1.0, ref.abs().max().item() / 4), (err, mode, M, N, K)


@pytest.mark.parametrize("mode", ["parity", "fast"])
def test_linear_residual_and_gelu(gu, mode):
    M, N, K = 333, 256, 1024
    a, w, b, r = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=K ** -0.5), rnd(N, seed=6), rnd(M, N, seed=7)
    out = gu.linear(a, w, b, resid=r, mode=mode)
    assert (out - (F.linear(a, w, b) + r)).abs().max().item() <= TOL[mode]
    out = gu.linear(a, w, b, gelu=True, mode=mode)
    assert (out - F.gelu(F.linear(a, w, b))).abs().max().item() <= TOL[mode]
    out = gu.linear(a, w, None, mode=mode)
    assert (out - F.linear(a, w)).abs().max().item() <= TOL[mode]


def test_linear_is_exact_on_small_integers(gu):
    """A = I-like / small-integer operands are exactly representable in bf16: the MFMA fragment
    layout (row/col maps, asymmetric B) must then reproduce the product bit-for-bit."""
    M, N, K = 192, 160, 128
    g = torch.Generator().manual_seed(9)
    a = torch.randint(-4, 5, (M, K), generator=g).float()
    w = torch.randint(-4, 5, (N, K), generator=g).float()
    for mode in ("fast", "parity"):
        out = gu.linear(a, w, None, mode=mode)
        assert torch.equal(out, a @ w.t()), mode
    eye = torch.eye(128)
    wasym = torch.arange(128 * 128, dtype=torch.float32).reshape(128, 128) % 251
    assert torch.equal(gu.linear(eye, wasym, None, mode="fast"), wasym.t())


def ref_attention(qkv, H):
    B, N, _ = qkv.shape
    q, k, v = qkv.reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    a = ((q * 0.125) @ k.transpose(-2, -1)).softmax(-1)
    return (a @ v).transpose(1, 2).reshape(B, N, H * 64)


@pytest.mark.parametrize("mode", ["parity", "fast"])
@pytest.mark.parametrize("B,N,H", [(1, 64, 1), (2, 40, 2), (1, 96, 1), (1, 97, 1), (1, 128, 2), (2, 200, 3), (1, 792, 2), (1, 785, 1), (1, 1568, 1), (1, 3168, 1),
                                   (1, 6272, 1), (1, 6336, 2), (8, 792, 1), (2, 1568, 4)])  # L/4 decoder, IMU decoder (padded), XCD-mapped grids (B*H % 8 == 0)
def test_attention_matches_dense_softmax(gu, mode, B, N, H):
    qkv = rnd(B, N, 3 * H * 64, seed=N)
    ref = ref_attention(qkv, H)
    out = gu.attention(qkv, H, mode=mode)
    assert torch.isfinite(out).all()
    err = (out - ref).abs().max().item()
    assert err <= (3e-4 if mode == "parity" else 3e-2), (err, mode, B, N, H)


def test_attention_online_softmax_rescale_branch(gu):
    """Force the running max to jump late (guide §5.4 rule 26): one key far above the rest in the
    LAST tile, and one in the first tile, for a subset of queries."""
    B, N, H = 1, 300, 1
    qkv = rnd(B, N, 192, seed=5)
    q = qkv[0, :, :64]
    k = qkv[0, :, 64:128]
    k[290] = q[7] * 6.0   # spike for query 7 in the last tile
    k[3] = q[100] * 6.0   # spike for query 100 in the first tile
    ref = ref_attention(qkv, H)
    for mode in ("parity", "fast"):
        out = gu.attention(qkv, H, mode=mode)
        err = (out - ref).abs().max().item()
        assert err <= (5e-4 if mode == "parity" else 5e-2), (mode, err)


def test_attention_golden_block(gu):
    """Reference `Attention`/`Block` tile at real width (tests/golden/block_768.npz)."""
    from collections import OrderedDict

    from counterfactualworldmodels_amd import config as C, synthetic as S

    g = np.load(os.path.join(GOLDEN, "block_768.npz"))
    shapes = OrderedDict()
    C._block_schema("", 768, 3072, shapes)
    W = {k: torch.from_numpy(S.synthetic_tensor("golden_block." + k, shp, int(g["seed"]))) for k, shp in shapes.items()}
    x = torch.from_numpy(g["x"])
    B, N, D = x.shape
    h1 = gu.layernorm(x.reshape(-1, D), W["norm1.weight"], W["norm1.bias"]).reshape(B, N, D)
    assert (h1 - torch.from_numpy(g["norm1"])).abs().max().item() <= 2e-6
    bias = torch.cat([W["attn.q_bias"], torch.zeros(D), W["attn.v_bias"]])
    qkv = gu.linear(h1.reshape(-1, D), W["attn.qkv.weight"], bias).reshape(B, N, 3 * D)
    o = gu.attention(qkv, 12)
    a = gu.linear(o.reshape(-1, D), W["attn.proj.weight"], W["attn.proj.bias"]).reshape(B, N, D)
    assert (a - torch.from_numpy(g["attn"])).abs().max().item() <= 2e-4
    # second half of Block.forward (VideoMAE/utils.py:150-152): x1 = x + attn; mlp = fc2(gelu(fc1(LN2 x1))); y = x1 + mlp
    x1 = gu.linear(o.reshape(-1, D), W["attn.proj.weight"], W["attn.proj.bias"], resid=x.reshape(-1, D))
    h2 = gu.layernorm(x1, W["norm2.weight"], W["norm2.bias"])
    f1 = gu.linear(h2, W["mlp.fc1.weight"], W["mlp.fc1.bias"], gelu=True)
    mlp = gu.linear(f1, W["mlp.fc2.weight"], W["mlp.fc2.bias"]).reshape(B, N, D)
    assert (mlp - torch.from_numpy(g["mlp"])).abs().max().item() <= 2e-4
    y = gu.linear(f1, W["mlp.fc2.weight"], W["mlp.fc2.bias"], resid=x1).reshape(B, N, D)
    assert (y - torch.from_numpy(g["y"])).abs().max().item() <= 2e-4


@pytest.mark.parametrize("D", [128, 192, 384, 512, 768, 1024])
def test_layernorm(gu, D):
    x = rnd(37, D, seed=D) * 3 + 1
    g_, b_ = 1 + 0.1 * rnd(D, seed=1), 0.1 * rnd(D, seed=2)
    ref = F.layer_norm(x, (D,), g_, b_, 1e-6)
    out = gu.layernorm(x, g_, b_)
    assert (out - ref).abs().max().item() <= 5e-6


def test_mask_to_perm_bit_exact_and_errors(gu):
    g = torch.Generator().manual_seed(0)
    for (B, Nt, nv) in [(3, 32, 20), (2, 1568, 792), (4, 6272, 3168), (1, 300, 1), (2, 257, 256)]:
        mask = torch.ones(B, Nt, dtype=torch.bool)
        for b in range(B):
            mask[b, torch.randperm(Nt, generator=g)[:nv]] = False
        perm = gu.mask_to_perm(mask, nv)
        for b in range(B):
            vis = torch.where(~mask[b])[0]
            msk = torch.where(mask[b])[0]
            assert torch.equal(perm[b].long(), torch.cat([vis, msk]))
    ragged = torch.ones(2, 64, dtype=torch.bool)
    ragged[0, :10] = False
    ragged[1, :11] = False
    with pytest.raises(_lib.CwmHipError) as e:
        gu.mask_to_perm(ragged, 10)
    assert e.value.code == _lib.ERR_MASK


def test_mask_row_counts_and_flip_picks_bit_exact(gu):
    """The two entry points behind `RectangularizeMasks` on device masks (cwm_mask_row_counts, cwm_mask_flip_picks) against numpy: counts of ragged rows,
    and picks applied to the row as it was before the call -- many picks per row, both directions, a row with no pick, row lengths that are not a multiple
    of the workgroup, the first and the last candidate, rows left untouched."""
    lib = _lib.get_lib()
    d = gu.dev()
    g = torch.Generator().manual_seed(5)
    for (B, Nt) in [(5, 40), (3, 257), (8, 1568), (4, 6336), (2, 16384)]:
        mask = torch.rand(B, Nt, generator=g) < 0.6
        mask[0, :] = True          # a fully masked row
        md = mask.to(d)
        counts = torch.empty(B, dtype=torch.int32, device=d)
        _lib.check(lib.cwm_mask_row_counts(md.data_ptr(), B, Nt, counts.data_ptr(), gu.stream()))
        assert torch.equal(counts.cpu().long(), mask.sum(1))
        ref = mask.clone().numpy()
        rows, offsets, to_value, picks = [], [0], [], []
        for b in range(B):
            if b == 1 and B > 2:
                continue            # untouched row
            to = b % 2              # even rows: un-mask masked tokens; odd rows: mask visible ones
            cand = np.flatnonzero(ref[b] != bool(to))
            if cand.size == 0:
                continue
            n_pick = [1, cand.size, min(7, cand.size), cand.size // 2 + 1][b % 4]
            k = torch.randperm(cand.size, generator=g)[:n_pick]
            if b % 4 == 2:
                k[0], k[-1] = 0, cand.size - 1    # the first and the last candidate
                k = torch.unique(k)
            ref[b, cand[k.numpy()]] = bool(to)
            rows.append(b); to_value.append(to); picks.append(k.to(torch.int32)); offsets.append(offsets[-1] + k.numel())
        table = torch.cat([torch.tensor([len(rows)] + rows + offsets + to_value, dtype=torch.int32)] + picks).to(d)
        _lib.check(lib.cwm_mask_flip_picks(md.data_ptr(), B, Nt, table.data_ptr(), len(rows), gu.stream()))
        assert np.array_equal(md.cpu().numpy(), ref), (B, Nt)
    with pytest.raises(_lib.CwmHipError):
        big = torch.zeros(1, 16385, dtype=torch.bool, device=d)
        _lib.check(lib.cwm_mask_flip_picks(big.data_ptr(), 1, 16385, table.data_ptr(), 1, gu.stream()))


def test_unembed_bit_exact_golden(gu):
    g = np.load(os.path.join(GOLDEN, "index_ops.npz"))
    lib = _lib.get_lib()
    d = gu.dev()
    y, x, m = (torch.from_numpy(g[k]).to(d) for k in ("unembed_y", "unembed_x", "unembed_mask"))
    out = torch.empty_like(x)
    B, T, Cc, H, W = x.shape
    n_vis = m.shape[1] - int(m[0].sum())
    _lib.check(lib.cwm_unembed(y.data_ptr(), x.data_ptr(), m.data_ptr(), B, T, Cc, H, W, 8, n_vis, out.data_ptr(), gu.stream()))
    assert np.array_equal(out.cpu().numpy(), g["unembed_video"])


@pytest.mark.parametrize("tile", [1, 4, 6])
def test_gemm_tile_configurations_agree(gud, tile):
    """All output-tile configurations of the GEMM (128x128, 256x256 8-phase, 8-phase rounds + 128x128 remainder rows) give the same result."""
    lib = gud.lib
    try:
        _lib.check(lib.cwm_debug_set(b"gemm_tile", tile), lib)
        for mode in ("parity", "fast"):
            for (M, N, K) in [(300, 768, 192), (1000, 1152, 384), (77, 48, 512), (700, 256, 64), (513, 384, 1536)]:
                a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
                out = gud.linear(a, w, b, mode=mode)
                err = (out - F.linear(a, w, b)).abs().max().item()
                assert err <= TOL[mode], (tile, mode, M, N, K, err)
        g = torch.Generator().manual_seed(9)
        a = torch.randint(-4, 5, (300, 128), generator=g).float()
        w = torch.randint(-4, 5, (272, 128), generator=g).float()
        assert torch.equal(gud.linear(a, w, None, mode="parity"), a @ w.t())
        assert torch.equal(gud.linear(a, w, None, mode="fast"), a @ w.t())
    finally:
        _lib.check(lib.cwm_debug_set(b"gemm_tile", 0), lib)


def test_gemm_mixed_tiling_is_bitwise_equal_to_simple_kernel(gud):
    """Mixed tiling (whole rounds of 256x256 8-phase tiles + a remainder of 128x128 tiles, split by rows): bit-identical to the
    128x128 kernel on a shape where the split happens (297 big-tile slots -> 255 + remainder), with bias + residual."""
    lib = gud.lib
    try:
        for mode in ("parity", "fast"):
            for (M, N, K) in [(25344, 768, 192), (20000, 512, 128)]:
                a, w, b, r = rnd(M, K, seed=41), rnd(N, K, seed=42, scale=K ** -0.5), rnd(N, seed=43), rnd(M, N, seed=44)
                _lib.check(lib.cwm_debug_set(b"gemm_tile", 1), lib)
                ref = gud.linear(a, w, b, resid=r, mode=mode)
                _lib.check(lib.cwm_debug_set(b"gemm_tile", 6), lib)
                out = gud.linear(a, w, b, resid=r, mode=mode)
                assert torch.equal(out, ref), (mode, M, N, K, (out - ref).abs().max().item())
    finally:
        _lib.check(lib.cwm_debug_set(b"gemm_tile", 0), lib)


def test_gemm_8phase_is_bitwise_equal_to_simple_kernel_under_repetition(gud):
    """Race screen for the 8-phase GEMM (LDS-DMA in flight across raw barriers, counted vmcnt): every accumulator
    sees the same product sequence as in the one-barrier-per-tile kernel, so the outputs must be bit-identical --
    for 1, 2, 3 and many K tiles, ragged M / N, repeated launches (a half-tile read before it landed, or re-staged
    before it was read, shows up as a mismatch)."""
    lib = gud.lib
    shapes = [(256, 256, 64), (300, 272, 128), (1000, 1152, 192), (513, 512, 384), (2049, 768, 768), (4096, 1024, 3072)]
    try:
        _lib.check(lib.cwm_debug_set(b"gemm_debug", 32), lib)  # (no split-K in the reference kernel: a split re-associates the fp32 sums)
        for mode in ("parity", "fast"):
            for (M, N, K) in shapes:
                a, w, b = rnd(M, K, seed=11), rnd(N, K, seed=12, scale=K ** -0.5), rnd(N, seed=13)
                _lib.check(lib.cwm_debug_set(b"gemm_tile", 1), lib)
                ref = gud.linear(a, w, b, mode=mode)
                for tile in (4,):   # 256x256 tiles (half-width instance for a last column tile of <= 128 columns)
                    _lib.check(lib.cwm_debug_set(b"gemm_tile", tile), lib)
                    for rep in range(8):
                        out = gud.linear(a, w, b, mode=mode)
                        assert torch.equal(out, ref), (mode, tile, M, N, K, rep, (out - ref).abs().max().item())
                    gl = gud.linear(a, w, b, gelu=True, mode=mode)   # (bf16 + GELU epilogue: the direct form with permuted W rows)
                    _lib.check(lib.cwm_debug_set(b"gemm_tile", 1), lib)
                    assert torch.equal(gl, gud.linear(a, w, b, gelu=True, mode=mode)), (mode, tile, M, N, K)
    finally:
        _lib.check(lib.cwm_debug_set(b"gemm_tile", 0), lib)
        _lib.check(lib.cwm_debug_set(b"gemm_debug", 0), lib)


def test_gemm_deep_ring_and_split_k_small_launches(gud):
    """Launches with fewer 128x128 tiles than CUs (batch 1) run the 4-stage-ring kernel (8 waves, counted vmcnt) and, for long K, split K
    over the idle CUs with a last-arriver reduction in fixed part order: bit-identical to the double-buffered kernel without a split,
    deterministic under repetition with it (a stale slab or a re-staged tile read too early shows up as a mismatch), every epilogue."""
    lib = gud.lib
    shapes = [(792, 768, 3072), (1568, 384, 1536), (792, 2304, 768), (300, 272, 2048), (130, 48, 4096), (1000, 256, 64), (3168, 768, 768), (40, 384, 768)]
    try:
        for mode in ("parity", "fast"):
            for (M, N, K) in shapes:
                a, w, b, r = rnd(M, K, seed=31), rnd(N, K, seed=32, scale=K ** -0.5), rnd(N, seed=33), rnd(M, N, seed=34)
                _lib.check(lib.cwm_debug_set(b"gemm_tile", 1), lib)
                _lib.check(lib.cwm_debug_set(b"gemm_debug", 4), lib)          # double-buffered kernel, no deep ring, no split
                ref = (gud.linear(a, w, b, resid=r, mode=mode), gud.linear(a, w, b, gelu=True, mode=mode))
                _lib.check(lib.cwm_debug_set(b"gemm_debug", 32), lib)         # deep ring (8 waves), no split: same product sequence
                for rep in range(3):
                    out = (gud.linear(a, w, b, resid=r, mode=mode), gud.linear(a, w, b, gelu=True, mode=mode))
                    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), (mode, M, N, K, rep)
                # (these shapes leave half of the CUs without a 128x128 tile, so the deep ring above ran its 64x128 tile; bit 8 keeps 128 rows)
                for bits, what in ((32 + 256, "128-row tiles, 8 waves"),):
                    _lib.check(lib.cwm_debug_set(b"gemm_debug", bits), lib)
                    out = (gud.linear(a, w, b, resid=r, mode=mode), gud.linear(a, w, b, gelu=True, mode=mode))
                    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), (mode, M, N, K, what)
                _lib.check(lib.cwm_debug_set(b"gemm_debug", 0), lib)          # default: + split-K where the heuristic takes it
                first = (gud.linear(a, w, b, resid=r, mode=mode), gud.linear(a, w, b, gelu=True, mode=mode))
                for rep in range(6):
                    out = (gud.linear(a, w, b, resid=r, mode=mode), gud.linear(a, w, b, gelu=True, mode=mode))
                    assert torch.equal(out[0], first[0]) and torch.equal(out[1], first[1]), (mode, M, N, K, rep, "split-K not deterministic")
                assert (first[0] - ref[0]).abs().max().item() <= (2e-5 if mode == "parity" else 2e-3), (mode, M, N, K)
                assert (first[0] - (F.linear(a, w, b) + r)).abs().max().item() <= TOL[mode]
    finally:
        _lib.check(lib.cwm_debug_set(b"gemm_tile", 0), lib)
        _lib.check(lib.cwm_debug_set(b"gemm_debug", 0), lib)


@pytest.mark.parametrize("tile", [1, 4])
def test_gemm_epilogue_forms_are_bitwise_equal(gud, tile):
    """The three epilogue forms only re-route the stores: the per-fragment form of round 1 (`gemm_staged` 0), the LDS-staged form (row
    table + 64x32 pieces written back as full row segments; `gemm_direct` 0) and the direct form (16-byte stores straight from the
    accumulators, W tile staged with permuted rows; `gemm_direct` 1 = bf16 outputs only, the default, 2 = fp32 outputs too) must give
    bit-identical outputs, for fp32 (+bias, +residual), GELU and plain outputs, ragged M / N and both modes."""
    lib = gud.lib
    cases = [(300, 272, 128), (1000, 1152, 192), (77, 48, 512), (513, 400, 384), (2049, 768, 768)]
    try:
        _lib.check(lib.cwm_debug_set(b"gemm_tile", tile), lib)
        for mode in ("parity", "fast"):
            for (M, N, K) in cases:
                a, w, b, r = rnd(M, K, seed=21), rnd(N, K, seed=22, scale=K ** -0.5), rnd(N, seed=23), rnd(M, N, seed=24)
                outs = []
                for staged, direct in ((0, 0), (1, 0), (1, 1), (1, 2)):
                    _lib.check(lib.cwm_debug_set(b"gemm_staged", staged), lib)
                    _lib.check(lib.cwm_debug_set(b"gemm_direct", direct), lib)
                    outs.append((gud.linear(a, w, b, mode=mode), gud.linear(a, w, b, resid=r, mode=mode), gud.linear(a, w, b, gelu=True, mode=mode),
                                 gud.linear(a, w, None, mode=mode)))
                for other in outs[1:]:
                    for o0, o1 in zip(outs[0], other):
                        assert torch.equal(o0, o1), (tile, mode, M, N, K)
                assert (outs[1][1] - (F.linear(a, w, b) + r)).abs().max().item() <= TOL[mode]
    finally:
        _lib.check(lib.cwm_debug_set(b"gemm_staged", 1), lib)
        _lib.check(lib.cwm_debug_set(b"gemm_direct", 1), lib)
        _lib.check(lib.cwm_debug_set(b"gemm_tile", 0), lib)


@pytest.mark.parametrize("kern", [3])
def test_attention_variants_are_bitwise_equal_to_4wave_kernel(gud, kern):
    """Race screen for the software-pipelined attention kernel (3: LDS-DMA K/V slots, S of tile t+1 interleaved with the softmax of
    tile t): per-wave arithmetic is that of the 4-wave kernel, so outputs must be bit-identical -- one tile, two tiles, odd and even
    tile counts, ragged tails, idle waves, the online-softmax rescale branch, repeated launches."""
    lib = gud.lib
    # (the last two fill the chip: more than one workgroup per CU)
    cases = [(2, 40, 1), (1, 64, 2), (2, 100, 1), (3, 129, 2), (1, 192, 2), (2, 300, 3), (2, 792, 12), (1, 1568, 6), (1, 1000, 2), (1, 3200, 1),
             (8, 792, 12), (6, 1568, 6)]
    try:
        # (the key-split schedule of a ragged last query tile, attention_tail.h, re-associates the key sum and exists in the 4-wave
        # workgroups only: this cross-check runs every tile on the regular schedule; the split has its own test below.  Likewise the key-split
        # tail round of the pipelined kernel, "attn_ksplit")
        _lib.check(lib.cwm_debug_set(b"attn_tail", 0), lib)
        _lib.check(lib.cwm_debug_set(b"attn_ksplit", 0), lib)
        for mode in ("parity", "fast"):
            for (B, N, H) in cases:
                qkv = rnd(B, N, 3 * H * 64, seed=N + 1)
                if N >= 300:
                    qkv[0, N - 5, H * 64:H * 64 + 64] = qkv[0, 7, :64] * 6.0  # late spike: the running max jumps in the last tile
                _lib.check(lib.cwm_debug_set(b"attn_kernel", 1), lib)
                ref = gud.attention(qkv, H, mode=mode)
                _lib.check(lib.cwm_debug_set(b"attn_kernel", kern), lib)
                for rep in range(4):
                    out = gud.attention(qkv, H, mode=mode)
                    assert torch.equal(out, ref), (mode, B, N, H, rep, (out - ref).abs().max().item())
    finally:
        _lib.check(lib.cwm_debug_set(b"attn_kernel", 0), lib)
        _lib.check(lib.cwm_debug_set(b"attn_tail", 1), lib)
        _lib.check(lib.cwm_debug_set(b"attn_ksplit", 1), lib)


def test_attention_key_split_of_the_ragged_last_tile(gud):
    """attention_tail.h: a last query tile of at most 32 rows (N = 129, 785, 792, 1568 ...) splits the KEYS of its tiles over the
    workgroup's four waves and merges four (max, sum, O) partials.  Against the regular schedule of the same kernel the rows of that
    tile may differ by the rounding of P (each wave has its own running maximum) and the order of the fp32 key sum, every other row must be bit-identical; the 4-wave and the pipelined kernel
    share the split code (bit-identical to each other); odd and even key-tile counts, a masked last key tile, a late spike in the last
    tile and one in the first, chip-filling grids; and the dense-softmax reference within the usual tolerance."""
    lib = gud.lib
    cases = [(2, 129, 1), (1, 160, 2), (2, 785, 1), (2, 792, 12), (1, 1568, 6), (8, 792, 12), (1, 897, 2), (1, 3104, 1)]
    try:
        _lib.check(lib.cwm_debug_set(b"attn_ksplit", 0), lib)   # (the other key split, of a whole tail ROUND, would take over where this one is switched off)
        for mode in ("parity", "fast"):
            for (B, N, H) in cases:
                qkv = rnd(B, N, 3 * H * 64, seed=N + 3)
                qkv[0, N - 2, H * 64:H * 64 + 64] = qkv[0, N - 1, :64] * 6.0   # late spike for the LAST query (a tail row)
                qkv[0, 3, H * 64:H * 64 + 64] = qkv[0, N - 3, :64] * 6.0       # early spike for another tail row
                tail0 = (N // 128) * 128 if N % 128 else N
                outs = {}
                for kern in (1, 3):
                    _lib.check(lib.cwm_debug_set(b"attn_kernel", kern), lib)
                    for tail in (0, 1):
                        _lib.check(lib.cwm_debug_set(b"attn_tail", tail), lib)
                        outs[kern, tail] = gud.attention(qkv, H, mode=mode)
                assert torch.equal(outs[1, 1], outs[3, 1]), (mode, B, N, H)
                assert torch.equal(gud.attention(qkv, H, mode=mode), outs[3, 1])           # deterministic
                assert torch.equal(outs[3, 1][:, :tail0], outs[3, 0][:, :tail0]), (mode, B, N, H)   # full tiles untouched
                d = (outs[3, 1][:, tail0:] - outs[3, 0][:, tail0:]).abs().max().item() if tail0 < N else 0.0
                # (every wave of the split exponentiates against its OWN running maximum, so the split-bf16 rounding of P -- 2^-17 relative -- falls on
                # other values than in the regular schedule: differences of that order times |v|, not only a re-associated sum)
                assert d <= (1e-4 if mode == "parity" else 1e-2), (mode, B, N, H, d)
                if N % 128 and N % 128 <= 32 and N > 128:
                    assert d > 0 or mode == "fast" or N == 129   # the split really ran (a re-associated sum seldom reproduces every bit)
                err = (outs[3, 1] - ref_attention(qkv, H)).abs().max().item()
                assert err <= (5e-4 if mode == "parity" else 5e-2), (mode, B, N, H, err)
    finally:
        _lib.check(lib.cwm_debug_set(b"attn_kernel", 0), lib)
        _lib.check(lib.cwm_debug_set(b"attn_tail", 1), lib)
        _lib.check(lib.cwm_debug_set(b"attn_ksplit", 1), lib)


def test_attention_key_split_tail_round(gud):
    """attention_pipe.hip, "attn_ksplit": when the last round of workgroups would fill at most half of the chip's slots, its work items
    are cut into key ranges (one workgroup each, partial (O, max, sum) records merged by attention_combine_kernel).  Rows of the split
    items may differ from the unsplit schedule by the fp32 re-association of the key sum and by where the rounding of P falls; every
    other row is bit-identical; both agree with the dense-softmax reference.  Shapes: 621 items = 1 round + 109 (3 key ranges of 6 tiles,
    a 76-row last query tile with an idle wave), 576 items = 1 round + 64 (2 ranges), and the ViT-L/4 decoder launch itself (3136 items =
    6 rounds + 64, 8 ranges) -- on / off only, plus one head against the dense reference."""
    lib = gud.lib
    try:
        for mode in ("parity", "fast"):
            for (B, N, H) in [(23, 1100, 3), (9, 1024, 8)]:
                qkv = rnd(B, N, 3 * H * 64, seed=N + 5)
                qkv[0, N - 5, H * 64:H * 64 + 64] = qkv[0, 7, :64] * 6.0  # late spike: the last key range holds the maximum of query 7
                if mode == "fast":
                    _lib.check(lib.cwm_debug_set(b"attn_kernel", 3), lib)       # (fast mode takes the pipelined kernel from 2048 tokens on)
                outs = []
                for ks in (0, 1):
                    _lib.check(lib.cwm_debug_set(b"attn_ksplit", ks), lib)
                    outs.append(gud.attention(qkv, H, mode=mode))
                assert torch.equal(gud.attention(qkv, H, mode=mode), outs[1])            # deterministic
                d = (outs[0] - outs[1]).abs()
                assert d.max().item() <= (1e-4 if mode == "parity" else 1e-2), (mode, B, N, H, d.max().item())
                changed = (d.amax(-1) > 0).sum().item()                                 # the split really ran, on rows of the last round's items only
                rem = (-(-N // 128) * B * H) % 512                                        # work items of the last round (512 slots on MI355X)
                assert 0 < changed <= rem * 128, (mode, B, N, H, changed)
                err = (outs[1] - ref_attention(qkv, H)).abs().max().item()
                assert err <= (5e-4 if mode == "parity" else 5e-2), (mode, B, N, H, err)
        _lib.check(lib.cwm_debug_set(b"attn_kernel", 0), lib)
        B, N, H = 8, 6272, 8
        qkv = rnd(B, N, 3 * H * 64, seed=77)
        for mode in ("parity", "fast"):
            outs = []
            for ks in (0, 1):
                _lib.check(lib.cwm_debug_set(b"attn_ksplit", ks), lib)
                outs.append(gud.attention(qkv, H, mode=mode))
            d = (outs[0] - outs[1]).abs()
            assert d.max().item() <= (1e-4 if mode == "parity" else 1e-2), (mode, d.max().item())
            assert 0 < (d.amax(-1) > 0).sum().item() <= 64 * 128
            one = qkv[3:4].reshape(1, N, 3, H, 64)[:, :, :, 5].reshape(1, N, 192)       # batch 3, head 5 through the dense reference
            ref = ref_attention(one, 1)
            err = (outs[1][3:4, :, 5 * 64:6 * 64] - ref).abs().max().item()
            assert err <= (5e-4 if mode == "parity" else 5e-2), (mode, err)
    finally:
        _lib.check(lib.cwm_debug_set(b"attn_ksplit", 1), lib)
        _lib.check(lib.cwm_debug_set(b"attn_kernel", 0), lib)
