"""A/B of GEMM kernel variants on one box: python tools/mb_variants.py  (tile,debug) pairs from VARIANTS env: e.g. "1:0,1:4,4:0,0:0" """
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from counterfactualworldmodels_amd import _lib  # noqa: E402
from tools.microbench import GEMM_SHAPES, MID_SHAPES, L4_SHAPES, B1_SHAPES  # noqa: E402

torch.cuda.init()
lib = _lib.get_lib()
us = C.c_double()
variants = [tuple(int(x) for x in v.split(":")) for v in os.environ.get("VARIANTS", "0:0,1:0,1:4,4:0").split(",")]
shapes = {"b8": GEMM_SHAPES, "mid": MID_SHAPES, "l4": L4_SHAPES, "b1": B1_SHAPES}[os.environ.get("SHAPES", "b8")]
modes = os.environ.get("MODES", "parity").split(",")
for name, M, N, K, epi in shapes:
    for mode in modes:
        cells = []
        for tile, dbg in variants:
            _lib.check(lib.cwm_debug_set(b"gemm_tile", tile))
            _lib.check(lib.cwm_debug_set(b"gemm_debug", dbg))
            _lib.check(lib.cwm_bench_gemm(M, N, K, _lib.mode_id(mode), epi, 30, C.byref(us)))
            cells.append("t%d/d%d %7.1f us %6.1f TF" % (tile, dbg, us.value, 2.0 * M * N * K / us.value / 1e6))
        print("%-14s %-6s M=%d N=%d K=%d  %s" % (name, mode, M, N, K, " | ".join(cells)), flush=True)
_lib.check(lib.cwm_debug_set(b"gemm_tile", 0))
_lib.check(lib.cwm_debug_set(b"gemm_debug", 0))
