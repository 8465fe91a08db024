"""ctypes binding of libcwm_hip.so (C ABI: include/cwm_hip.h).

There is no CPU fallback: if the library cannot be loaded every entry point raises.
`import torch` must happen before the library is loaded so that both share one HIP runtime
(libamdhip64.so.7 is resolved by soname to the copy torch already mapped).
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

import torch  # noqa: F401  (loads libamdhip64 first; see module docstring)

from . import build as _build

MODE_FAST = 1
MODE_PARITY = 2
ERR_INVALID = -1
ERR_HIP = -2
ERR_MASK = -3
KCLASS_GEMM = 0
KCLASS_ATTENTION = 1
KCLASS_GEMM_WIDE = 2    # 256x256 8-phase kernel launches (booked together with KCLASS_GEMM)
KCLASS_GEMM_NARROW = 3  # 128x128 kernel launches
KCLASS_LAYERNORM, KCLASS_PATCH_GATHER, KCLASS_FILL_MASK, KCLASS_UNEMBED = 4, 5, 6, 7  # HBM-bound: `total_flops` = algorithmic BYTES
KCLASS_CROSS_ATTN, KCLASS_SMALL_ATTN = 8, 9  # IMU-conditioned model

_MODES = {"fast": MODE_FAST, "parity": MODE_PARITY, MODE_FAST: MODE_FAST, MODE_PARITY: MODE_PARITY}


def mode_id(mode) -> int:
    try:
        return _MODES[mode]
    except KeyError:
        raise ValueError("mode must be 'fast' or 'parity', got %r" % (mode,))


class CwmConfig(C.Structure):
    _fields_ = [
        ("img_h", C.c_int32),
        ("img_w", C.c_int32),
        ("patch", C.c_int32),
        ("num_frames", C.c_int32),
        ("in_chans", C.c_int32),
        ("enc_dim", C.c_int32),
        ("enc_depth", C.c_int32),
        ("enc_heads", C.c_int32),
        ("dec_dim", C.c_int32),
        ("dec_depth", C.c_int32),
        ("dec_heads", C.c_int32),
        ("mlp_ratio", C.c_int32),
        ("ln_eps", C.c_float),
    ]


class CwmForwardArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),  # = C.sizeof(CwmForwardArgs): new_forward_args() sets it
        ("x_dev", C.c_void_p),
        ("x_stride_b", C.c_int64),
        ("x_stride_c", C.c_int64),
        ("x_stride_t", C.c_int64),
        ("normalize", C.c_int32),
        ("mask_dev", C.c_void_p),
        ("batch", C.c_int32),
        ("n_vis", C.c_int32),
        ("y_tokens_dev", C.c_void_p),
        ("y_video_dev", C.c_void_p),
        ("xraw_dev", C.c_void_p),
        ("mode", C.c_int32),
        ("check", C.c_int32),
        ("stream", C.c_void_p),
    ]


class CwmConjConfig(C.Structure):
    _fields_ = [
        ("main", CwmConfig),
        ("main_max_pad", C.c_int32),
        ("ctx_in_chans", C.c_int32),
        ("ctx_seq_len", C.c_int32),
        ("ctx_tubelet", C.c_int32),
        ("ctx_enc_dim", C.c_int32),
        ("ctx_dec_dim", C.c_int32),
        ("ctx_enc_heads", C.c_int32),
        ("ctx_dec_heads", C.c_int32),
        ("ctx_max_pad", C.c_int32),
        ("n_enc_cross", C.c_int32),
        ("enc_cross", C.c_int32 * 16),
        ("n_dec_cross", C.c_int32),
        ("dec_cross", C.c_int32 * 16),
        ("cross_heads", C.c_int32),
        ("cross_mlp_ratio", C.c_int32),
    ]


class CwmConjForwardArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("x_dev", C.c_void_p),
        ("x_stride_b", C.c_int64),
        ("x_stride_c", C.c_int64),
        ("x_stride_t", C.c_int64),
        ("normalize", C.c_int32),
        ("mask_dev", C.c_void_p),
        ("batch", C.c_int32),
        ("n_vis_max", C.c_int32),
        ("ctx_dev", C.c_void_p),
        ("ctx_mask_dev", C.c_void_p),
        ("n_vis_ctx_max", C.c_int32),
        ("y_tokens_dev", C.c_void_p),
        ("mode", C.c_int32),
        ("check", C.c_int32),
        ("stream", C.c_void_p),
        ("y_ctx_tokens_dev", C.c_void_p),
    ]


def new_forward_args() -> CwmForwardArgs:
    a = CwmForwardArgs()
    a.struct_size = C.sizeof(CwmForwardArgs)
    return a


def new_conj_forward_args() -> CwmConjForwardArgs:
    a = CwmConjForwardArgs()
    a.struct_size = C.sizeof(CwmConjForwardArgs)
    return a


class CwmKernelStats(C.Structure):
    _fields_ = [("launches", C.c_int64), ("total_ms", C.c_double), ("total_flops", C.c_double)]


# name -> (restype, argtypes); must list every symbol declared in include/cwm_hip.h
SIGNATURES = {
    "cwm_model_create": (C.c_int, [C.POINTER(CwmConfig), C.POINTER(C.c_void_p)]),
    "cwm_model_destroy": (None, [C.c_void_p]),
    "cwm_model_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "cwm_model_missing_weights": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "cwm_forward": (C.c_int, [C.c_void_p, C.POINTER(CwmForwardArgs)]),
    "cwm_model_set_lanes": (C.c_int, [C.c_void_p, C.c_int]),
    "cwm_model_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "cwm_conj_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "cwm_conj_create": (C.c_int, [C.POINTER(CwmConjConfig), C.POINTER(C.c_void_p)]),
    "cwm_conj_destroy": (None, [C.c_void_p]),
    "cwm_conj_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "cwm_conj_missing_weights": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "cwm_conj_forward": (C.c_int, [C.c_void_p, C.POINTER(CwmConjForwardArgs)]),
    "cwm_conj_set_lanes": (C.c_int, [C.c_void_p, C.c_int]),
    "cwm_conj_timing_enable": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "cwm_conj_timing_collect": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(CwmKernelStats)]),
    "cwm_timing_enable": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "cwm_timing_collect": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(CwmKernelStats)]),
    "cwm_split_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cwm_linear": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    ),
    "cwm_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cwm_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "cwm_mask_to_perm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cwm_unembed": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    ),
    "cwm_mask_row_counts": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cwm_mask_flip_picks": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "cwm_prompt_table_expand": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 4),
    "cwm_shift_prompts": (
        C.c_int,
        [C.c_void_p] + [C.c_int] * 9 + [C.c_void_p] * 6,
    ),
    "cwm_flow_features": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p]),
    "cwm_flow_cov": (C.c_int, [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p] * 4),
    "cwm_flow_transform_work_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "cwm_flow_transform": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "cwm_flow_motion_work_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "cwm_flow_motion_sum": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)] + [C.c_int] * 6 + [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cwm_flow_map_finish": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, C.c_void_p]),
    "cwm_comm_load": (C.c_int, [C.c_char_p]),
    "cwm_comm_version": (C.c_int, []),
    "cwm_comm_unique_id": (C.c_int, [C.c_void_p]),
    "cwm_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "cwm_comm_destroy": (None, [C.c_void_p]),
    "cwm_comm_rank": (C.c_int, [C.c_void_p]),
    "cwm_comm_size": (C.c_int, [C.c_void_p]),
    "cwm_broadcast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "cwm_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cwm_allgatherv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p]),
    "cwm_allreduce_sum_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cwm_last_error": (C.c_char_p, []),
    "cwm_version": (C.c_char_p, []),
    "cwm_source_hash": (C.c_char_p, []),
    "cwm_compiler_version": (C.c_char_p, []),
}
# ... and every symbol of include/cwm_hip_dev.h: only libcwm_hip_dev.so (get_dev_lib) has these
DEV_SIGNATURES = {
    "cwm_gemm_tile_override": (C.c_int, [C.c_int] * 6),
    "cwm_bench_gemm": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "cwm_bench_attention": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "cwm_debug_set": (C.c_int, [C.c_char_p, C.c_int]),
    "cwm_debug_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
}

# the keys cwm_model_set_option / cwm_conj_set_option know (include/cwm_hip.h; csrc/engine.hip tuning_field)
OPTION_KEYS = ("gemm_tile", "gemm_debug", "gemm_staged", "gemm_direct", "attn_kernel", "attn_remap", "attn_tail", "attn_ksplit", "prune_last_block", "index_fused",
               "min_lane_rows", "conj_ctx_stream", "conj_attn")
GEMM_DEBUG_TIMING_ONLY = 1 | 2 | 8  # ablation bits that make a forward return wrong outputs: refused by the production setters (development library: cwm_debug_set)


def validate_option(key: str, value: int) -> None:
    """What the library's setters would refuse, raised BEFORE a model stores the option for a handle it does not have yet (a bad key remembered there
    would fail every later handle creation with an unrelated-looking error)."""
    if key not in OPTION_KEYS:
        raise CwmHipError(-1, "unknown option %s (known: %s)" % (key, ", ".join(OPTION_KEYS)))
    if key == "gemm_debug" and int(value) & GEMM_DEBUG_TIMING_ONLY:
        raise CwmHipError(-1, "gemm_debug bits 1, 2 and 8 are timing-only ablations (wrong outputs): development library only (cwm_debug_set)")


_lib: Optional[C.CDLL] = None
_dev_lib: Optional[C.CDLL] = None
_lock = threading.Lock()


class CwmHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


def library_path() -> str:
    return os.environ.get("CWM_HIP_LIB", _build.LIB_PATH)


def get_lib() -> C.CDLL:
    """Load (building if absent and hipcc is available) libcwm_hip.so.  Raises if impossible."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.exists(path):
            try:
                _build.build_library()
            except Exception as e:  # no silent fallback
                raise RuntimeError(
                    "libcwm_hip.so is missing at %s and could not be built (%s). "
                    "Run `python -m counterfactualworldmodels_amd.build`." % (path, e)
                )
        lib = _bind(path)
        if path == _build.LIB_PATH and os.path.isdir(_build.CSRC):
            # a stale in-tree library (sources changed after it was built) is rebuilt, never silently bound
            want = _build.source_hash()
            if lib.cwm_source_hash().decode() != want:
                try:
                    _build.build_library(force=False)
                except Exception as e:
                    raise RuntimeError("libcwm_hip.so at %s was built from other sources (%s != %s) and could not be rebuilt: %s"
                                       % (path, lib.cwm_source_hash().decode(), want, e))
                # the dynamic loader returns the already-mapped object for a path it knows: unmap the stale library first (nothing was
                # created through it yet), then map the rebuilt file
                import _ctypes

                _ctypes.dlclose(lib._handle)
                lib = _bind(path)
                if lib.cwm_source_hash().decode() != want:
                    raise RuntimeError("libcwm_hip.so still does not match the sources after a rebuild")
        _warn_if_unlinted_compiler(lib)
        _lib = lib
        return lib


def _warn_if_unlinted_compiler(lib) -> None:
    """The kernels carry hand-counted `s_waitcnt` instructions that are only as right as the ISA the compiler emitted around them (tools/asm_lds_lint.py);
    a library compiled by another hipcc than the one the lint last passed on (csrc/LINT_PASSED.json) says so once, at load time."""
    try:
        rec = _build.lint_record().get("hipcc")
        comp = lib.cwm_compiler_version().decode()
    except Exception:  # a library from before the entry point existed / no record in an installed copy
        return
    if rec and comp and rec != comp:
        import warnings

        warnings.warn("libcwm_hip.so was compiled by `%s`, but the ISA lint of its hand-counted waits last passed on `%s`: run `python tools/asm_lds_lint.py --record` "
                      "(and the GPU soak, tests/test_soak_gpu.py) before trusting this build" % (comp, rec), RuntimeWarning, stacklevel=3)


def dev_library_path() -> str:
    p = library_path()
    return os.environ.get("CWM_HIP_DEV_LIB", _build.DEV_LIB_PATH if p == _build.LIB_PATH else p[:-3] + "_dev.so")


def get_dev_lib() -> C.CDLL:
    """The development library (libcwm_hip_dev.so: the production objects + csrc/dev.hip; include/cwm_hip_dev.h) -- for tools/ and for the tests that
    cross-check kernel variants.  A SEPARATE shared object with its own state: handles created through it are its own; `cwm_debug_set` acts on the
    calling thread's options inside it and never on the production library."""
    global _dev_lib
    if _dev_lib is not None:
        return _dev_lib
    get_lib()  # builds / rebuilds both libraries if the sources changed
    with _lock:
        if _dev_lib is None:
            path = dev_library_path()
            if not os.path.exists(path):
                raise RuntimeError("libcwm_hip_dev.so is missing at %s: run `python -m counterfactualworldmodels_amd.build`" % path)
            lib = _bind(path, dict(SIGNATURES, **DEV_SIGNATURES))
            if path == _build.DEV_LIB_PATH and os.path.isdir(_build.CSRC) and lib.cwm_source_hash().decode() != _build.source_hash():
                raise RuntimeError("libcwm_hip_dev.so at %s was built from other sources than the tree holds: run `python -m counterfactualworldmodels_amd.build`" % path)
            _dev_lib = lib
    return _dev_lib


def _bind(path: str, signatures=None) -> C.CDLL:
    lib = C.CDLL(path)
    for name, (res, args) in (signatures or SIGNATURES).items():
        try:
            fn = getattr(lib, name)  # a missing symbol fails loudly
        except AttributeError:
            if name == "cwm_source_hash":  # a library from before the hash existed: report it as stale
                lib.cwm_source_hash = lambda: b"pre-hash"
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    return lib


def check(rc: int, lib: Optional[C.CDLL] = None) -> None:
    """Raise for a negative return code; `lib` = the library the call went through (default: the production one; a call through the development
    library that fails leaves its message there -- the thread-local error text is per shared object)."""
    if rc != 0:
        msg = (lib or get_lib()).cwm_last_error()
        if not msg and lib is None and _dev_lib is not None:
            msg = _dev_lib.cwm_last_error()
        raise CwmHipError(rc, (msg.decode() if msg else "") + " [cwm_hip rc=%d]" % rc)


def require_gpu() -> None:
    if not torch.cuda.is_available():
        raise RuntimeError(
            "counterfactualworldmodels_amd runs its predictor on an AMD GPU through libcwm_hip.so; "
            "no HIP device is visible and there is no CPU fallback."
        )


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def current_stream_handle(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream
