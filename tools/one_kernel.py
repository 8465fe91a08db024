"""Launch one kernel configuration a few times (for rocprofv3 --pmc runs).
    python tools/one_kernel.py gemm M N K mode epi variant | attn B H N mode"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from counterfactualworldmodels_amd import _lib  # noqa: E402

torch.cuda.init()
lib = _lib.get_dev_lib()
us = C.c_double()
if sys.argv[1] == "gemm":
    M, N, K = (int(v) for v in sys.argv[2:5])
    mode, epi, variant = sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
    _lib.check(lib.cwm_bench_gemm(M, N, K, _lib.mode_id(mode), epi, 3, C.byref(us)))
else:
    B, H, N = (int(v) for v in sys.argv[2:5])
    _lib.check(lib.cwm_debug_set(b"attn_kernel", int(os.environ.get("ATTN_KERNEL", "0"))))
    _lib.check(lib.cwm_bench_attention(B, H, N, _lib.mode_id(sys.argv[5]), 3, C.byref(us)))
print(sys.argv[1:], "%.1f us" % us.value)
