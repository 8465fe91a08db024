"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/cwm_hip.h
declares (no compute calls here)."""
import ctypes
import os
import re

from counterfactualworldmodels_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="cwm_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cwm_[a-z0-9_]+)\s*\(", src)))


def exported_symbols(path):
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if line.strip())


def test_library_builds_and_exports_all_declared_symbols():
    path = build.build_library()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert sorted(_lib.SIGNATURES) == names  # the ctypes binding covers exactly the header


def test_production_library_exports_the_documented_c_abi_and_nothing_else():
    """`nm -D libcwm_hip.so` = the entry points include/cwm_hip.h declares: no C++ internals, no kernel handles, and none of the development entry
    points (switches, micro-benchmarks, per-shape tile overrides: include/cwm_hip_dev.h), which live in libcwm_hip_dev.so only."""
    import shutil

    if not shutil.which("nm"):
        import pytest

        pytest.skip("nm not available")
    build.build_library()
    prod, dev = exported_symbols(build.LIB_PATH), exported_symbols(build.DEV_LIB_PATH)
    assert prod == declared_symbols(), sorted(set(prod) ^ set(declared_symbols()))
    dev_only = [n for n in declared_symbols("cwm_hip_dev.h") if n not in declared_symbols()]
    assert sorted(_lib.DEV_SIGNATURES) == dev_only
    assert dev == sorted(prod + dev_only)
    assert not any(n in prod for n in ("cwm_debug_set", "cwm_bench_gemm", "cwm_bench_attention", "cwm_gemm_tile_override"))
    d = _lib.get_dev_lib()  # loads beside the production library (separate state), binds every symbol of both headers
    assert d.cwm_source_hash() == _lib.get_lib().cwm_source_hash()
    assert d.cwm_debug_set(b"attn_kernel", 3) == 0 and d.cwm_debug_set(b"attn_kernel", 0) == 0   # (thread-local options of the dev object: no GPU needed)
    assert d.cwm_debug_set(b"no_such_switch", 1) != 0 and b"unknown key" in d.cwm_last_error()


def test_header_cites_reference_and_version_string():
    src = open(os.path.join(ROOT, "include", "cwm_hip.h")).read()
    assert "prediction.py:419-422" in src and "vmae.py:539-560" in src
    lib = _lib.get_lib()
    assert lib.cwm_version().decode().startswith("cwm_hip")
    assert lib.cwm_last_error() is not None


def test_struct_layout_matches_header():
    # 12 int32 + 1 float
    assert ctypes.sizeof(_lib.CwmConfig) == 13 * 4
    assert ctypes.sizeof(_lib.CwmKernelStats) == 24
    # pointers/int64 aligned to 8
    # struct_size first (padded to 8), then as before
    assert ctypes.sizeof(_lib.CwmForwardArgs) == 8 + 8 * 4 + 4 + 4 + 8 + 4 + 4 + 8 * 3 + 4 + 4 + 8
    assert _lib.new_forward_args().struct_size == ctypes.sizeof(_lib.CwmForwardArgs)
    assert _lib.new_conj_forward_args().struct_size == ctypes.sizeof(_lib.CwmConjForwardArgs)
    assert _lib.get_lib().cwm_compiler_version().decode() == build.hipcc_version().replace('"', "'")


def test_development_switches_are_per_thread_not_per_process():
    """The only switchboard left is the development library's, and it is THREAD-local (kernels.h thread_tuning): a value set on one thread is not
    seen on another, every thread starts from the defaults, and the production library has no such entry point at all.  (Per-MODEL independence
    needs a device: tests/test_model_gpu.py::test_options_are_per_model_handle.)  No GPU needed: the options are host state."""
    import threading

    d = _lib.get_dev_lib()

    def get(key):
        v = ctypes.c_int(-123)
        assert d.cwm_debug_get(key, ctypes.byref(v)) == 0
        return v.value

    defaults = {b"attn_kernel": 0, b"gemm_tile": 0, b"gemm_direct": 1, b"attn_remap": 1, b"prune_last_block": 1, b"index_fused": 1, b"min_lane_rows": 0}
    assert {k: get(k) for k in defaults} == defaults
    seen, barrier = {}, threading.Barrier(2)

    def worker(name, kernel, tile):
        assert {k: get(k) for k in defaults} == defaults          # a fresh thread: the defaults, whatever other threads did
        assert d.cwm_debug_set(b"attn_kernel", kernel) == 0 and d.cwm_debug_set(b"gemm_tile", tile) == 0
        barrier.wait()                                             # both threads have written ...
        seen[name] = (get(b"attn_kernel"), get(b"gemm_tile"))      # ... and each still reads its own

    ts = [threading.Thread(target=worker, args=("a", 1, 4)), threading.Thread(target=worker, args=("b", 3, 6))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert seen == {"a": (1, 4), "b": (3, 6)}
    assert {k: get(k) for k in defaults} == defaults               # the main thread never saw either
    v = ctypes.c_int()
    assert d.cwm_debug_get(b"no_such_switch", ctypes.byref(v)) != 0
    assert not hasattr(ctypes.CDLL(build.LIB_PATH), "cwm_debug_get")
