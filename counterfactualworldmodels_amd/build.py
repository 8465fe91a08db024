"""Build libcwm_hip.so (hipcc, gfx950) in-tree.  `python -m counterfactualworldmodels_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcwm_hip.so")
SOURCES = ["gemm.hip", "gemm_sk.hip", "attention.hip", "attention_pipe.hip", "elementwise.hip", "conj_kernels.hip", "flowstats.hip", "engine.hip", "model.hip", "conj_model.hip"]
HEADERS = ["common.h", "kernels.h", "gemm_device.h", "attention_device.h", "engine.h", os.path.join("..", "..", "include", "cwm_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [
        _hipcc(),
        "--offload-arch=gfx950",
        "-O3",
        "-std=c++17",
        "-fPIC",
        "-shared",
        "-Wall",
        "-Wno-unused-function",
    ]
    cmd += os.environ.get("CWM_HIPCC_EXTRA", "").split()  # e.g. -DCWM_ATTN_PROF (phase timers in attention_pipe.hip)
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
