"""K sweep of one GEMM shape through the C ABI (development tool, GPU box): time = fixed + slope * K tells the
per-tile prologue/epilogue cost from the main-loop rate.   python tools/ksweep.py M N epi [variants]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from counterfactualworldmodels_amd import _lib  # noqa: E402


def main():
    M, N, epi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    variants = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "1,3,4").split(",")]
    torch.cuda.init()
    lib = _lib.get_dev_lib()
    us = C.c_double()
    for mode in ("fast", "parity"):
        for v in variants:
            _lib.check(lib.cwm_debug_set(b"gemm_tile", v))
            row = []
            for K in (64, 128, 256, 512, 768, 1536, 3072):
                best = 1e30
                for _ in range(3):
                    _lib.check(lib.cwm_bench_gemm(M, N, K, _lib.mode_id(mode), epi, 20, C.byref(us)))
                    best = min(best, us.value)
                row.append("K%d %.1f" % (K, best))
            print("M=%d N=%d epi=%d %-6s t%d  %s" % (M, N, epi, mode, v, " | ".join(row)), flush=True)
    _lib.check(lib.cwm_debug_set(b"gemm_tile", 0))


if __name__ == "__main__":
    main()
