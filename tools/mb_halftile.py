"""Half-width column tiles of the 8-phase GEMM kernel ("gemm_debug" bit 1024 = off: round-4 rule and kernel): the GEMM shapes with N mod 256 = 128, full and half
(two-lane) batch, parity and fast.   python tools/mb_halftile.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import _lib
torch.cuda.init()
lib = _lib.get_dev_lib(); us = C.c_double()
SHAPES = [("b8.dec.fc2", 50176, 384, 1536, 0), ("b8.dec.fc2.half", 25088, 384, 1536, 0), ("b8.e2d", 25344, 384, 768, 0), ("b8.dec.qkv", 50176, 1152, 384, 3), ("b8.dec.qkv.half", 25088, 1152, 384, 3),
          ("b8.dec.proj", 50176, 384, 384, 0), ("imu.dec.fc2", 101376, 384, 1536, 0), ("imu.dec.fc2.half", 50688, 384, 1536, 0), ("imu.dec.qkv", 101376, 1152, 384, 3)]
for name, M, N, K, epi in SHAPES:
    for mode in ("parity", "fast"):
        res = {}
        for rep in range(3):
            for dbg in (1024, 0):
                _lib.check(lib.cwm_debug_set(b"gemm_debug", dbg), lib)
                _lib.check(lib.cwm_bench_gemm(M, N, K, _lib.mode_id(mode), epi, 20, C.byref(us)), lib)
                res.setdefault(dbg, []).append(us.value)
        a, b = min(res[1024]), min(res[0])
        print("%-18s M=%6d N=%4d K=%4d %-6s  round-4 %7.1f us  half tiles %7.1f us  (%+.1f %%)  %6.1f -> %6.1f TFLOP/s" % (name, M, N, K, mode, a, b, 100 * (b / a - 1), 2.0 * M * N * K / a / 1e6, 2.0 * M * N * K / b / 1e6), flush=True)
_lib.check(lib.cwm_debug_set(b"gemm_debug", 0), lib)
