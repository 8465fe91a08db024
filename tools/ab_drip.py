"""Probe (side build with -DCWM_DRIP_PROBE): cost of output stores issued inside the 8-phase main loop instead of in the epilogue.
    CWM_HIP_LIB=/path/to/side.so python tools/ab_drip.py
Columns: full kernel (direct epilogue) | epilogue without its global stores | the same + 1 / 2 / 3 store instructions per K tile inside the loop | no epilogue."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from counterfactualworldmodels_amd import _lib  # noqa: E402

torch.cuda.init()
lib = _lib.get_lib()
us = C.c_double()
SHAPES = [("b8.enc.qkv-like", 25344, 2304, 768, 2), ("b8.enc.fc1", 25344, 3072, 768, 1), ("b8.dec.qkv-like", 50176, 1152, 384, 2), ("b8.dec.fc1", 50176, 1536, 384, 1),
          ("l4.enc.fc1", 25344, 4096, 1024, 1), ("b8x16.enc.fc1", 12672, 3072, 768, 1)]
_lib.check(lib.cwm_debug_set(b"gemm_tile", 4))
_lib.check(lib.cwm_debug_set(b"gemm_direct", 1))
for name, M, N, K, epi in SHAPES:
    cells = []
    for dbg in (0, 1, 1 + 512, 1 + 2048, 1 + 1024, 2):
        _lib.check(lib.cwm_debug_set(b"gemm_debug", dbg))
        best = 1e30
        for _ in range(3):
            _lib.check(lib.cwm_bench_gemm(M, N, K, _lib.mode_id("parity"), epi, 20, C.byref(us)))
            best = min(best, us.value)
        cells.append("d%-4d %6.1f us" % (dbg, best))
    print("%-16s M=%d N=%d K=%d  %s" % (name, M, N, K, " | ".join(cells)), flush=True)
_lib.check(lib.cwm_debug_set(b"gemm_debug", 0))
_lib.check(lib.cwm_debug_set(b"gemm_tile", 0))
_lib.check(lib.cwm_debug_set(b"gemm_direct", 0))
