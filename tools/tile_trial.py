"""Every tile configuration (Tuning.gemm_tile 1: 128x128, 4: 256x256 8-phase, 6: 8-phase rounds + 128x128 remainder rows; 0: the rule's choice) on the GEMM shapes
where the vendor library beat the fast-mode kernels by more than 10 % (profiles/r4_library_yardstick.log), both modes, next to torch.matmul bf16 on the same box.

    python tools/tile_trial.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import _lib
torch.cuda.init()
lib = _lib.get_dev_lib(); us = C.c_double()
SHAPES = [("l4.enc.proj", 25344, 1024, 1024, 0), ("l4.enc.fc2", 25344, 1024, 4096, 0), ("l4.dec.fc2", 50176, 512, 2048, 0),
          ("l4.enc.proj.half", 12672, 1024, 1024, 0), ("l4.enc.fc2.half", 12672, 1024, 4096, 0), ("l4.dec.fc2.half", 25088, 512, 2048, 0)]
def t_lib(M, N, K, resid=False):
    """torch.matmul bf16 (plain bf16 output); resid: + what the model's epilogue also does -- bias and the fp32 residual stream updated in place
    (x += y + b: 8 bytes per element read-modify-write, which the library GEMM does not fuse)"""
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(M, N, device="cuda"); b = torch.randn(N, device="cuda")
    def step():
        y = torch.matmul(a, w.t())
        if resid:
            x.add_(y).add_(b)
    for _ in range(3): step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for name, M, N, K, epi in SHAPES:
    t_lib(M, N, K)
    base, full = t_lib(M, N, K), t_lib(M, N, K, True)
    row = ["%-18s M=%d N=%d K=%d  library bf16 %6.1f us, + bias + fp32 residual update %6.1f us" % (name, M, N, K, base, full)]
    for mode in ("fast", "parity"):
        _lib.check(lib.cwm_bench_gemm(M, N, K, _lib.mode_id(mode), epi, 10, C.byref(us)), lib)  # (warm-up: clocks)
        for tile in (0, 1, 4, 6):
            _lib.check(lib.cwm_debug_set(b"gemm_tile", tile), lib)
            best = 1e30
            for _ in range(2):
                _lib.check(lib.cwm_bench_gemm(M, N, K, _lib.mode_id(mode), epi, 20, C.byref(us)), lib)
                best = min(best, us.value)
            row.append("%s t%d %6.1f us%s" % (mode, tile, best, " (%.2fx lib, %.2fx lib + update)" % (best / base, best / full) if mode == "fast" else ""))
    print(" | ".join(row), flush=True)
_lib.check(lib.cwm_debug_set(b"gemm_tile", 0), lib)
