"""Epilogue ablation of the GEMM (development tool, GPU box): python tools/ablate.py M N epi tile"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from counterfactualworldmodels_amd import _lib  # noqa: E402

M, N, epi, tile = (int(v) for v in sys.argv[1:5])
torch.cuda.init()
lib = _lib.get_dev_lib()
us = C.c_double()
_lib.check(lib.cwm_debug_set(b"gemm_tile", tile))
for mode in ("fast", "parity"):
    for dbg in (0, 1, 2):
        _lib.check(lib.cwm_debug_set(b"gemm_debug", dbg))
        row = []
        for K in (64, 768, 3072):
            best = 1e30
            for _ in range(3):
                _lib.check(lib.cwm_bench_gemm(M, N, K, _lib.mode_id(mode), epi, 20, C.byref(us)))
                best = min(best, us.value)
            row.append("K%d %.1f" % (K, best))
        print("M=%d N=%d epi=%d t%d %-6s debug=%d  %s" % (M, N, epi, tile, mode, dbg, " | ".join(row)), flush=True)
_lib.check(lib.cwm_debug_set(b"gemm_debug", 0))
_lib.check(lib.cwm_debug_set(b"gemm_tile", 0))
