"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/cwm_hip.h
declares (no compute calls here)."""
import ctypes
import os
import re

from counterfactualworldmodels_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "cwm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cwm_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_all_declared_symbols():
    path = build.build_library()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert sorted(_lib.SIGNATURES) == names  # the ctypes binding covers exactly the header


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
    assert ctypes.sizeof(_lib.CwmForwardArgs) == 8 * 4 + 4 + 4 + 8 + 4 + 4 + 8 * 3 + 4 + 4 + 8
