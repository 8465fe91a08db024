"""Host-side logic of the package that needs no GPU: schema, factories/initialisation parity,
mask helpers, synthetic generators, error behaviour without a device."""
import os

import numpy as np
import pytest
import torch

from counterfactualworldmodels_amd import config as C
from counterfactualworldmodels_amd import masking, synthetic as S, vmae
from oracle import vmae_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TINY = C.VmaeConfig(name="tiny_8x8", img_size=(32, 32), patch=8, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128,
                    dec_depth=1, dec_heads=2)


def test_parameter_counts():
    # known-answer parameter counts (ipynb:244 for L/4)
    assert C.num_parameters(C.CONFIGS["base_8x8patch_2frames_1tube"]) == 92_661_312
    assert C.num_parameters(C.CONFIGS["large_4x4patch_2frames_1tube"]) == 340_709_936
    assert len(C.state_dict_schema(C.CONFIGS["base_8x8patch_2frames_1tube"])) == 218
    assert C.CONFIGS["base_8x8patch_2frames_1tube"].num_tokens == 1568
    assert C.CONFIGS["large_4x4patch_2frames_1tube"].num_tokens == 6272


def test_module_state_dict_matches_schema():
    m = vmae.PretrainVisionTransformer(TINY)
    sd = m.state_dict()
    schema = C.state_dict_schema(TINY)
    assert list(sd.keys()) == list(schema.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == schema[k], k
    assert m.patch_size == (1, 8, 8) and m.mask_size == (2, 4, 4) and m.num_frames == 2
    assert m.encoder.patch_embed.proj.kernel_size == (1, 8, 8)


@pytest.mark.parametrize("seed", [0, 5])
def test_constructor_rng_parity_with_reference(seed):
    """Same seed -> same freshly initialised parameters as the reference constructor (fixture
    captured from the reference by make_golden.py)."""
    g = np.load(os.path.join(GOLDEN, "init_tiny.npz"))
    torch.manual_seed(seed)
    m = vmae.PretrainVisionTransformer(TINY)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g[f"keys_{seed}"]]
    sums = np.array([v.double().sum().item() for v in sd.values()])
    abss = np.array([v.double().abs().sum().item() for v in sd.values()])
    assert np.allclose(sums, g[f"sums_{seed}"], rtol=0, atol=1e-9)
    assert np.allclose(abss, g[f"abs_{seed}"], rtol=0, atol=1e-9)


def test_factories_exist():
    for name in ("base_8x8patch_2frames_1tube", "base_16x16patch_2frames_1tube", "large_4x4patch_2frames_1tube"):
        assert callable(getattr(vmae, name))


def test_rectangularize_matches_oracle_and_mutates_in_place():
    g = np.load(os.path.join(GOLDEN, "index_ops.npz"))
    a = torch.from_numpy(g["rect_in"].copy())
    torch.manual_seed(int(g["rect_seed"]))
    out = masking.RectangularizeMasks("min")(a)
    assert np.array_equal(out.numpy(), g["rect_out"])
    assert np.array_equal(a.numpy(), g["rect_out"])  # in place, like the reference
    same = torch.from_numpy(g["rect_out"].copy())
    assert torch.equal(masking.RectangularizeMasks("min")(same.clone()), same)  # idempotent on rectangular input
    assert masking.RectangularizeMasks("full")(same).all()
    assert masking.RectangularizeMasks(None)(same) is same


def test_rectangularize_empty_and_ragged():
    r = masking.RectangularizeMasks("min")
    z = torch.zeros(3, 10, dtype=torch.bool)
    assert not r(z.clone()).any()
    ragged = torch.zeros(3, 10, dtype=torch.bool)
    ragged[0, :7] = True
    ragged[1, :2] = True
    ragged[2, :] = True
    out = r(ragged.clone())
    assert out.sum(-1).tolist() == [2, 2, 2]
    assert (out & ~ragged).sum() == 0  # only ever un-masks
    # the common count is kept for the predictor wrapper (it is the n_vis the library needs, read back in the same host sync)
    assert r.last_num_masked == 2
    for mode, want in (("max", 10), ("mean", 6)):
        rr = masking.RectangularizeMasks(mode)
        assert rr(ragged.clone()).sum(-1).tolist() == [want] * 3 and rr.last_num_masked == want
    for mode in ("full", None):
        rr = masking.RectangularizeMasks(mode)
        rr(ragged.clone())
        assert rr.last_num_masked is None  # nothing was counted: the wrapper falls back to asking the library


def test_upsample_masks():
    m = torch.tensor([[True, False], [False, True]])[None]
    up = masking.upsample_masks(m, (4, 4))
    assert up.shape == (1, 4, 4) and up[0, 0, 1] and not up[0, 0, 2] and up[0, 3, 3]
    assert masking.upsample_masks(up, (2, 2)).equal(m)


def test_synthetic_generators_are_stable():
    cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
    w = S.synthetic_tensor("encoder.blocks.0.attn.qkv.weight", (2304, 768), 0)
    assert w.dtype == np.float32 and abs(float(w.std()) - float(np.sqrt(2.0 / (2304 + 768))) ) < 1e-3
    x = S.synthetic_frames(1, cfg, 0)
    assert x.shape == (1, 2, 3, 224, 224) and 0 <= x.min() and x.max() < 1
    mk = S.synthetic_masks(4, cfg, 8, 0)
    assert mk.shape == (4, 1568) and not mk[:, :784].any() and ((~mk).sum(-1) == 792).all()
    mk = S.synthetic_masks(2, C.CONFIGS["large_4x4patch_2frames_1tube"], 32, 0, clump=2)
    assert ((~mk).sum(-1) == 3168).all()
    p = S.synthetic_prompts(256, cfg, 0)
    assert p.shape == (256, 4) and (np.abs(p[:, 2:]).max() <= 3) and (np.abs(p[:, 2:]).sum(-1) > 0).all()


def test_forward_without_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = vmae.PretrainVisionTransformer(TINY)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 2, 32, 32), torch.zeros(1, 32, dtype=torch.bool))
    with pytest.raises(RuntimeError):
        m.decoder(torch.zeros(1))


def test_oracle_spec_matches_config():
    for name, cfg in C.CONFIGS.items():
        s = O.SPECS[name]
        for f in ("patch", "enc_dim", "enc_depth", "enc_heads", "dec_dim", "dec_depth", "dec_heads", "mlp_ratio"):
            assert getattr(s, f) == getattr(cfg, f)


def test_weight_sync_signature_sees_every_way_a_parameter_can_change():
    """vmae.WeightSync (no GPU needed: only the host-side change detector).  Cases from the round-2 review: `p.data = t` (new storage,
    same version counter), a dtype conversion of a SUBMODULE (never reaches the top-level `_apply`), a Parameter replaced with setattr."""
    import torch
    from counterfactualworldmodels_amd import config as C, vmae

    cfg = C.VmaeConfig(name="tiny_8x8", img_size=(32, 32), patch=8, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128, dec_depth=1, dec_heads=2)
    m = vmae.PretrainVisionTransformer(cfg)
    assert not m._params_unchanged()           # nothing uploaded yet
    m._remember_params()
    assert m._params_unchanged()
    assert len(m._plist) == len(m.state_dict())
    w = m.decoder.head.weight
    v0 = w._version
    w.data = w.data.clone()                    # storage pointer changes, version counter does not
    assert w._version == v0 and not m._params_unchanged()
    m._remember_params()
    m.encoder.double()                         # submodule conversion
    assert not m._params_unchanged()
    m.encoder.float()
    m._remember_params()
    assert m._params_unchanged()
    m.decoder.head.bias = torch.nn.Parameter(torch.zeros_like(m.decoder.head.bias))   # replaced object
    assert not m._params_unchanged()
    m._remember_params()
    with torch.no_grad():
        m.mask_token.add_(1.0)                 # in-place: version counter
    assert not m._params_unchanged()
    m._remember_params()
    m.load_state_dict(m.state_dict())          # post hook forgets the list
    assert not m._params_unchanged()


def test_bench_helpers_on_cpu(tmp_path, monkeypatch):
    """bench.py's host-side helpers that do not need a GPU: the committed-profile lookup returns (bytes, mfma busy, tag) for every
    workload / mode (Nones when the profile holds no entry for the command), the executed-work fraction, the GPU count without a HIP call,
    and `--gpus N` without N GPUs / a launcher-flag mismatch exit 2 -- deterministically, whatever the box has (the children see at
    most one device)."""
    import json
    import subprocess
    import sys

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench

    for wl, mode in (("base8", "parity"), ("base8", "fast"), ("large4", "parity"), ("imu4", "parity"), ("nope", "parity")):
        got = bench.pmc_profile(wl, mode, "cwm::gemm8p_kernel<2>")
        assert isinstance(got, tuple) and len(got) == 3
        if wl == "nope":
            assert got == (None, None, None)
    # the committed summary carries every bench workload in parity mode (no null traffic on the three lines)
    with open(os.path.join(ROOT, "profiles", "pmc_summary_latest.json")) as f:
        entries = json.load(f)["entries"]
    for wl in ("base8", "large4", "imu4"):
        e = entries["%s/parity" % wl]
        assert e["hbm_bytes_per_launch"] and all(v > 0 for v in e["hbm_bytes_per_launch"].values()), wl
        assert e["mfma_busy"] and all(0.0 < v <= 1.0 for v in e["mfma_busy"].values()), wl
    pf = bench.profile_fields("base8", "parity", "cwm::gemm8p_kernel<2>", 0.18)
    assert abs(pf["executed_frac"] - 0.54) < 1e-12 and set(pf) == {"traffic", "mfma_busy", "executed_frac", "from_profile"}
    assert bench.profile_fields("base8", "fast", "x", 0.3)["executed_frac"] == 0.3
    st = {"launches": 4, "total_ms": 0.1, "total_flops": 4 * 155.7e6}
    e = bench.edge_kernels(lambda kc: st if kc == bench._lib.KCLASS_LAYERNORM else {"launches": 0, "total_ms": 0.0, "total_flops": 0.0})
    assert set(e) == {"layernorm_kernel"} and abs(e["layernorm_kernel"]["achieved"] - 6228.0) < 1.0 and e["layernorm_kernel"]["unit"] == "GB/s"
    # GPU count from the KFD topology (no HIP call); the *_VISIBLE_DEVICES lists cap it
    n = bench.count_gpus_without_hip()
    assert n is None or n >= 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    n1 = bench.count_gpus_without_hip()
    assert n1 is None or n1 <= 1
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="0", ROCR_VISIBLE_DEVICES="0", CUDA_VISIBLE_DEVICES="0")
    env.pop("WORLD_SIZE", None)
    env.pop("CWM_BENCH_ONE_DEVICE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env)
    assert r.returncode == 2 and "refusing" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=dict(env, WORLD_SIZE="1"))
    assert r.returncode == 2 and "launcher started 1" in r.stderr


def _lint():
    import shutil
    import sys

    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import asm_lds_lint

    return asm_lds_lint


def test_hand_placed_lds_waits_cover_every_asm_read():
    """tools/asm_lds_lint.py on the gfx950 ISA of attention_pipe.hip (hipcc cross-compiles without a GPU): no instruction reads a register that an inline-asm
    `ds_read_b64_tr_b16` writes before the hand-counted `s_waitcnt lgkmcnt` that covers the read, and no branch or label sits between such a read and its
    wait.  hipcc does not know these registers are written late and may copy them at a control-flow merge; round 4 shipped such a copy for a few hours (5 % of
    the ViT-L/4 batch-8 forwards wrong in one sample, two lanes only)."""
    asm_lds_lint = _lint()
    text = asm_lds_lint.compile_isa("attention_pipe.hip")
    assert len(__import__("re").findall(r";;#ASMSTART\n\s*ds_read_b64_tr_b16", text)) > 100          # the lint looks at the code it is meant for
    hits = asm_lds_lint.lint_isa(text)
    assert not hits, hits[:5]
    A, E = ";;#ASMSTART\n", ";;#ASMEND\n"
    rd = lambda r, a="v1": A + "\tds_read_b64_tr_b16 %s, %s\n" % (r, a) + E
    wt = lambda n: A + "\ts_waitcnt lgkmcnt(%d)\n" % n + E
    # the rules fire on the hazards: a use between the read and its wait ...
    bad = "_Zk:\n" + rd("v[4:5]") + "\tv_mov_b64_e32 v[8:9], v[4:5]\n" + wt(0) + "\ts_endpgm\n"
    assert len(asm_lds_lint.lint_isa(bad)) == 2
    # ... a branch or a label (control-flow merge) with a read outstanding ...
    bad = "_Zk:\n" + rd("v[4:5]") + "\ts_cbranch_vccnz .LBB0_2\n\tv_mfma_f32_32x32x16_bf16 v[10:25], v[30:33], v[34:37], v[10:25]\n.LBB0_2:\n" + wt(0) + "\ts_endpgm\n"
    assert [h[3] for h in asm_lds_lint.lint_isa(bad)] == [-1, -1]
    # ... and a compiler-issued LDS read behind the asm ones shifts what a counted wait retires
    bad = "_Zk:\n" + rd("v[4:5]") + "\tds_read_b128 v[40:43], v2\n" + wt(1) + "\tv_mov_b64_e32 v[8:9], v[4:5]\n" + wt(0) + "\ts_endpgm\n"
    assert asm_lds_lint.lint_isa(bad) == []          # (one younger operation may stay outstanding: the asm read itself has returned)
    bad = "_Zk:\n" + rd("v[4:5]") + rd("v[6:7]") + "\tds_read_b128 v[40:43], v2\n" + wt(2) + "\tv_mov_b64_e32 v[8:9], v[6:7]\n" + wt(0) + "\ts_endpgm\n"
    assert len(asm_lds_lint.lint_isa(bad)) == 2      # lgkmcnt(2) leaves v[6:7] and the b128 in flight
    ok = "_Zk:\n" + rd("v[4:5]") + rd("v[6:7]") + wt(1) + "\tv_mov_b64_e32 v[8:9], v[4:5]\n" + wt(0) + "\ts_cbranch_scc1 .LBB0_1\n.LBB0_1:\n\ts_endpgm\n"
    assert asm_lds_lint.lint_isa(ok) == []
    # compiler-managed transposed reads (the builtin: no ASMSTART) are the compiler's business
    ok = "_Zk:\n\tds_read_b64_tr_b16 v[4:5], v1\n\ts_cbranch_scc1 .LBB0_1\n.LBB0_1:\n\ts_waitcnt lgkmcnt(0)\n\ts_endpgm\n"
    assert asm_lds_lint.lint_isa(ok) == []


def test_counted_vmcnt_waits_match_the_lds_dma_the_compiler_emitted():
    """The second family of hand-counted waits: LDS-DMA (`global_load_lds_dwordx4`) kept in flight across raw barriers and retired by `s_waitcnt vmcnt(N)`
    (gemm.hip: the 8-phase kernel's vmcnt(6), the deep-ring kernel's vmcnt(2 PER) / vmcnt(PER)).  N counts instructions, so the ISA of every such loop must
    hold exactly the pieces the source counted and no other vector-memory instruction (a spill, a hoisted or duplicated piece).  Fires on a synthetic fault of
    each kind; the compiler a build uses is compared with the one this lint last passed on (csrc/LINT_PASSED.json, build.check_lint_record)."""
    asm_lds_lint = _lint()
    from counterfactualworldmodels_amd import build

    text = asm_lds_lint.compile_isa("gemm.hip")
    import re

    assert len(re.findall(r";;#ASMSTART\n\s*s_waitcnt vmcnt\([1-9]", text)) >= 16   # 2 modes x 2 tile instances x (prologue + loop) of the 8-phase kernel, 2 per deep-ring kernel
    assert asm_lds_lint.lint_vmcnt(text) == []
    assert asm_lds_lint.lint_isa(text) == []
    # every kernel with a counted wait has a spec derived from its template arguments
    for name in re.findall(r"^(_Z\w+):", text, flags=re.M):
        if "gemm8p_kernel" in name:
            assert asm_lds_lint.vmcnt_spec(name)["loop"] == {6: (8, 8), 4: (6, 6)}   # full tile / half-width column tile
    first = text.index("_ZN3cwm13gemm8p_kernelILi2EEEvNS_10GemmParamsE:")
    loop = text.index("Loop Header", first)
    piece = text.index("global_load_lds_dwordx4", loop)
    eol = text.index("\n", piece)
    spill = text[:eol + 1] + "\tscratch_store_dword off, v1, off offset:4\n" + text[eol + 1:]
    assert any("vector-memory instruction inside a loop" in h[2] for h in asm_lds_lint.lint_vmcnt(spill))
    dup = text[:eol + 1] + "\t" + text[piece:eol + 1] + text[eol + 1:]
    assert any("9 LDS-DMA" in h[2] for h in asm_lds_lint.lint_vmcnt(dup))
    imm = text[:first] + text[first:].replace("s_waitcnt vmcnt(6)", "s_waitcnt vmcnt(7)")
    assert any("vmcnt(7) is not one the source places" in h[2] for h in asm_lds_lint.lint_vmcnt(imm))
    swapped = text[:first] + text[first:].replace("s_waitcnt vmcnt(4)", "s_waitcnt vmcnt(6)")   # the half-width tile instance with the full tile's count
    assert any("issues 6 LDS-DMA pieces" in h[2] for h in asm_lds_lint.lint_vmcnt(swapped))
    pro = text.index("global_load_lds_dwordx4", first)
    e2 = text.index("\n", pro)
    assert any("among the prologue" in h[2] for h in asm_lds_lint.lint_vmcnt(text[:e2 + 1] + "\tglobal_load_dword v1, v[2:3], off\n" + text[e2 + 1:]))
    # the record of the compiler the lint passed on is current (re-run `python tools/asm_lds_lint.py --record` after a toolchain bump)
    rec = build.lint_record()
    assert rec.get("hipcc") == build.hipcc_version(), "csrc/LINT_PASSED.json names another compiler: %r" % rec.get("hipcc")
    assert build.check_lint_record(verbose=False)


def test_option_keys_of_the_host_mirror_match_the_library_table():
    """`_lib.OPTION_KEYS` (what `model.set_option` accepts before a handle exists) must be the key table of `tuning_field` in csrc/engine.hip, and the timing-only
    `gemm_debug` bits the host refuses must be the ones `tuning_set_production` refuses: a key added on one side only would either be rejected by the mirror or
    poison a later handle creation."""
    import re

    from counterfactualworldmodels_amd import _lib, build

    src = open(os.path.join(build.CSRC, "engine.hip")).read()
    table = src[src.index("static const Field fields[] = {"):src.index("for (const Field& f : fields)")]
    keys = re.findall(r'\{"([a-z_]+)", &Tuning::', table)
    assert keys and sorted(keys) == sorted(_lib.OPTION_KEYS)
    m = re.search(r'strcmp\(key, "gemm_debug"\) && \(value & \(([0-9 |]+)\)\)', src)
    assert m and eval(m.group(1)) == _lib.GEMM_DEBUG_TIMING_ONLY
    _lib.validate_option("attn_kernel", 3)
    for bad in (("no_such_option", 1), ("gemm_debug", 8), ("gemm_debug", 32 | 2)):
        with pytest.raises(_lib.CwmHipError):
            _lib.validate_option(*bad)
    _lib.validate_option("gemm_debug", 32 | 1024)
