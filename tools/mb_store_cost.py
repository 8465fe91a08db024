import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from counterfactualworldmodels_amd import _lib
torch.cuda.init()
lib = _lib.get_dev_lib(); us = C.c_double()
_lib.check(lib.cwm_debug_set(b"gemm_tile", 4))
for (M, N, K, epi, name) in [(4096, 1024, 768, 3, "64 tiles (quarter chip) qkv-like"), (8192, 2048, 768, 3, "256 tiles (one round)"), (16384, 4096, 768, 3, "1024 tiles (4 rounds)"),
                             (4096, 1024, 768, 1, "64 tiles gelu"), (8192, 2048, 768, 1, "256 tiles gelu"), (16384, 4096, 768, 1, "1024 tiles gelu")]:
    if epi == 3:
        N = (N // 192) * 192  # qkv_dim multiple of 64*3
    cells = []
    for dbg in (0, 1, 2):
        _lib.check(lib.cwm_debug_set(b"gemm_debug", dbg))
        _lib.check(lib.cwm_bench_gemm(M, N, K, _lib.mode_id("parity"), epi, 30, C.byref(us)))
        cells.append(us.value)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    rounds = -(-tiles // 256)
    print("%-34s M=%d N=%d tiles %d: full %.1f us, no stores %.1f, no epilogue %.1f  -> per round: stores %.1f us, rest of epilogue %.1f us, main loop %.1f us"
          % (name, M, N, tiles, cells[0], cells[1], cells[2], (cells[0] - cells[1]) / rounds, (cells[1] - cells[2]) / rounds, cells[2] / rounds), flush=True)
_lib.check(lib.cwm_debug_set(b"gemm_debug", 0)); _lib.check(lib.cwm_debug_set(b"gemm_tile", 0))
