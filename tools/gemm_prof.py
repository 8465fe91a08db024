"""Per-workgroup timeline of the 8-phase GEMM (TILE=4, default) or the 128x128 kernel (TILE=1): shader clock under load, main loop vs
epilogue; needs a library built with CWM_HIPCC_EXTRA=-DCWM_GEMM_PROF.   [TILE=1] python tools/gemm_prof.py M N K mode epi"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from counterfactualworldmodels_amd import _lib  # noqa: E402

torch.cuda.init()
lib = _lib.get_dev_lib()
us = C.c_double()
M, N, K = (int(v) for v in sys.argv[1:4])
mode, epi = sys.argv[4], int(sys.argv[5])
TILE = int(os.environ.get("TILE", "4"))
BT = 256 if TILE == 4 else 128
_lib.check(lib.cwm_debug_set(b"gemm_tile", TILE))
_lib.check(lib.cwm_bench_gemm(M, N, K, _lib.mode_id(mode), epi, 5, C.byref(us)))
assert lib.cwm_debug_set(b"gemm_prof", 0) == 0, "library built without -DCWM_GEMM_PROF"
nblk = min(8192, ((M + BT - 1) // BT) * ((N + BT - 1) // BT))
r = np.fromfile("/tmp/gemm_blocks.bin", dtype=np.uint64).reshape(-1, 4)[:nblk].astype(np.int64)
t0 = r[:, 0].min()
start, end, cyc, main = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0, r[:, 2], r[:, 3]
dur = end - start
planes = 2 if mode == "parity" else 1
mfma = (BT * BT * K * (3 if planes == 2 else 1)) / (16 * 16 * 32) / 8 * 16  # MFMA pipe cycles per wave (8 waves, 16 cycles each)
print("%s: %.1f us (%.0f TF algorithmic); %d workgroups, span %.1f us" % (sys.argv[1:], us.value, 2.0 * M * N * K / us.value / 1e6, nblk, end.max()))
print("  workgroup duration us: median %.1f (min %.1f max %.1f); shader clock %.2f GHz; cycles: total %.0f main loop %.0f (%.0f %%), MFMA pipe busy in the main loop %.0f %%"
      % (np.median(dur), dur.min(), dur.max(), np.median(cyc / dur) / 1e3, np.median(cyc), np.median(main), 100 * np.median(main / cyc), 100 * mfma * (2 if TILE == 4 else 4) / np.median(main)))  # waves per SIMD sharing the pipe: 2 (one 256x256 workgroup) / 4 (two 128x128)
# how the workgroups of one CU overlap: at any instant, how many workgroups are inside their main loop (sampled over the launch)
ts = np.linspace(0, end.max(), 2000)
main_end = start + dur * (main / np.maximum(cyc, 1))
in_main = ((start[None, :] <= ts[:, None]) & (ts[:, None] < main_end[None, :])).sum(1)
in_epi = ((main_end[None, :] <= ts[:, None]) & (ts[:, None] < end[None, :])).sum(1)
print("  mean workgroups in the main loop %.0f, in the epilogue %.0f (of %d resident slots)" % (in_main.mean(), in_epi.mean(), 256 * (1 if TILE == 4 else 2)))
