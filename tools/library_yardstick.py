"""Vendor-library yardstick (tools only -- nothing of this is in the product path): the shapes of the predictor's GEMMs and of the
ViT-L/4 decoder attention through PyTorch's ROCm libraries (torch.matmul -> hipBLASLt / rocBLAS, F.scaled_dot_product_attention) next to
this library's kernels, same box, same process.  A stated baseline, not a target: the library GEMMs write ONE bf16 / fp32 output with no
bias / GELU / hi-lo split / head scatter, the kernels here include their fused epilogues.

    python tools/library_yardstick.py > profiles/r4_library_yardstick.log
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from counterfactualworldmodels_amd import _lib  # noqa: E402
from tools.microbench import GEMM_SHAPES, L4_SHAPES  # noqa: E402

torch.cuda.init()
lib = _lib.get_dev_lib()
us = C.c_double()
dev = torch.device("cuda:0")


def time_us(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / iters * 1e6)
    return best


def ours(M, N, K, mode, epi):
    best = 1e30
    for _ in range(3):
        _lib.check(lib.cwm_bench_gemm(M, N, K, _lib.mode_id(mode), epi, 20, C.byref(us)))
        best = min(best, us.value)
    return best


print("# GEMM: TFLOP/s = 2*M*N*K / time.  ours fast = one bf16 MFMA product, ours parity = three (split-bf16), both with the model's fused epilogue;")
print("# torch bf16 = torch.matmul(bf16, bf16) (one product, plain bf16 output), torch fp32 = torch.matmul(fp32, fp32) (the arithmetic the parity mode replaces;")
print("# MI355X fp32 matrix peak: 157 TFLOP/s)")
for name, M, N, K, epi in GEMM_SHAPES + L4_SHAPES:
    a16 = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w16 = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
    t_bf16 = time_us(lambda: torch.matmul(a16, w16.t()))
    a32, w32 = a16.float(), w16.float()
    t_f32 = time_us(lambda: torch.matmul(a32, w32.t()), iters=5, warm=1)
    del a32, w32
    t_fast, t_par = ours(M, N, K, "fast", epi), ours(M, N, K, "parity", epi)
    fl = 2.0 * M * N * K / 1e6
    print("%-12s M=%5d N=%4d K=%4d | ours fast %7.1f us %6.1f TF | torch bf16 %7.1f us %6.1f TF (ours/lib time %.2f) | ours parity %7.1f us %6.1f TF | torch fp32 %8.1f us %5.1f TF (ours/lib time %.2f)"
          % (name, M, N, K, t_fast, fl / t_fast, t_bf16, fl / t_bf16, t_fast / t_bf16, t_par, fl / t_par, t_f32, fl / t_f32, t_par / t_f32), flush=True)

print("# attention, ViT-L/4 decoder shape (B 8, H 8, N 6272, d 64) and ViT-B/8 encoder (B 32, H 12, N 792): TFLOP/s = 4*B*H*N*N*64 / time")
for B, H, N in ((8, 8, 6272), (32, 12, 792)):
    q = torch.randn(B, H, N, 64, device=dev, dtype=torch.bfloat16)
    k, v = torch.randn_like(q), torch.randn_like(q)
    fl = 4.0 * B * H * N * N * 64 / 1e6
    row = []
    try:
        t = time_us(lambda: F.scaled_dot_product_attention(q, k, v), iters=5, warm=2)
        row.append("torch SDPA bf16 %8.1f us %6.1f TF" % (t, fl / t))
    except Exception as e:  # noqa: BLE001
        row.append("torch SDPA bf16 failed: %s" % type(e).__name__)
    for mode in ("fast", "parity"):
        best = 1e30
        for _ in range(3):
            _lib.check(lib.cwm_bench_attention(B, H, N, _lib.mode_id(mode), 10, C.byref(us)))
            best = min(best, us.value)
        row.append("ours %s %8.1f us %6.1f TF" % (mode, best, fl / best))
    print("attn B=%d H=%d N=%d | %s" % (B, H, N, " | ".join(row)), flush=True)
