"""Per-GEMM cost of the LayerNorm-fold epilogues: every B/8 shape with its plain epilogue and with the fold form (epi + 10)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import _lib
from tools.microbench import GEMM_SHAPES, MID_SHAPES
torch.cuda.init()
lib = _lib.get_lib(); us = C.c_double()
for shapes in (GEMM_SHAPES, MID_SHAPES):
    for name, M, N, K, epi in shapes:
        row = []
        for e in (epi, epi + 10, epi, epi + 10):
            _lib.check(lib.cwm_bench_gemm(M, N, K, _lib.MODE_PARITY, e, 30, C.byref(us)))
            row.append("epi %2d %7.1f us" % (e, us.value))
        print("%-14s M=%d N=%d K=%d  %s" % (name, M, N, K, " | ".join(row)), flush=True)
