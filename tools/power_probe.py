"""Is the MFMA clock under load set by instantaneous or by time-averaged power?  The same GEMM launched back to back and with idle
gaps of 0.5x / 1x / 3x its own duration after every launch; prints the mean duration of the GEMM launches alone.
    python tools/power_probe.py   (GPU box)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from counterfactualworldmodels_amd import _lib  # noqa: E402

torch.cuda.init()
lib = _lib.get_dev_lib()
us = C.c_double()
for name, M, N, K, epi in [("enc.fc2", 25344, 768, 3072, 0), ("enc.qkv-like fc1", 25344, 3072, 768, 1), ("8192^3", 8192, 8192, 8192, 0)]:
    for mode in ("parity", "fast"):
        row = []
        base = None
        for gap_mult in (0.0, 0.5, 1.0, 3.0):
            gap = 0 if base is None else int(base * gap_mult)
            iters = 400 if M * N * K < 3e11 else 60
            _lib.check(lib.cwm_bench_gemm_gapped(M, N, K, _lib.mode_id(mode), epi, iters, gap, C.byref(us)))
            if base is None:
                base = us.value
            row.append("gap %5d us: %7.1f us" % (gap, us.value))
        print("%-18s %-6s %s" % (name, mode, " | ".join(row)), flush=True)
