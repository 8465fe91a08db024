"""Single-kernel micro-benchmarks through the C ABI (development tool, run on the GPU box).

    python tools/microbench.py gemm      # GEMM shapes of the B/8 batch-32 and L/4 batch-8 forward
    python tools/microbench.py attn
    python tools/microbench.py attn_sweep   # 4-wave vs software-pipelined kernel from 40 to 6336 tokens (the fast-mode selection rule of attention.hip)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from counterfactualworldmodels_amd import _lib  # noqa: E402

GEMM_SHAPES = [
    # (name, M, N, K, epi)
    ("b8.enc.qkv", 25344, 2304, 768, 3),
    ("b8.enc.proj", 25344, 768, 768, 0),
    ("b8.enc.fc1", 25344, 3072, 768, 1),
    ("b8.enc.fc2", 25344, 768, 3072, 0),
    ("b8.dec.qkv", 50176, 1152, 384, 3),
    ("b8.dec.proj", 50176, 384, 384, 0),
    ("b8.dec.fc1", 50176, 1536, 384, 1),
    ("b8.dec.fc2", 50176, 384, 1536, 0),
]
MID_SHAPES = [  # B/8 at batch 8 and 16 (encoder rows 792 B, decoder rows 1568 B)
    ("b8x8.enc.qkv", 6336, 2304, 768, 3),
    ("b8x8.enc.proj", 6336, 768, 768, 0),
    ("b8x8.enc.fc1", 6336, 3072, 768, 1),
    ("b8x8.enc.fc2", 6336, 768, 3072, 0),
    ("b8x8.dec.qkv", 12544, 1152, 384, 3),
    ("b8x8.dec.fc1", 12544, 1536, 384, 1),
    ("b8x16.enc.qkv", 12672, 2304, 768, 3),
    ("b8x16.enc.proj", 12672, 768, 768, 0),
    ("b8x16.enc.fc1", 12672, 3072, 768, 1),
    ("b8x16.enc.fc2", 12672, 768, 3072, 0),
    ("b8x16.dec.qkv", 25088, 1152, 384, 3),
    ("b8x16.dec.fc1", 25088, 1536, 384, 1),
]
L4_SHAPES = [
    ("l4.enc.qkv", 25344, 3072, 1024, 3),
    ("l4.enc.proj", 25344, 1024, 1024, 0),
    ("l4.enc.fc1", 25344, 4096, 1024, 1),
    ("l4.enc.fc2", 25344, 1024, 4096, 0),
    ("l4.dec.qkv", 50176, 1536, 512, 3),
    ("l4.dec.proj", 50176, 512, 512, 0),
    ("l4.dec.fc1", 50176, 2048, 512, 1),
    ("l4.dec.fc2", 50176, 512, 2048, 0),
]
B1_SHAPES = [  # ViT-B/8 at batch 1 (latency-bound launches: fewer tiles than CUs)
    ("b1.enc.qkv", 792, 2304, 768, 3),
    ("b1.enc.proj", 792, 768, 768, 0),
    ("b1.enc.fc1", 792, 3072, 768, 1),
    ("b1.enc.fc2", 792, 768, 3072, 0),
    ("b1.dec.qkv", 1568, 1152, 384, 3),
    ("b1.dec.proj", 1568, 384, 384, 0),
    ("b1.dec.fc1", 1568, 1536, 384, 1),
    ("b1.dec.fc2", 1568, 384, 1536, 0),
]
ATTN_SHAPES = [("b8.enc", 32, 12, 792), ("b8.dec", 32, 6, 1568), ("l4.enc", 8, 16, 3168), ("l4.dec", 8, 8, 6272)]


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    variants = [int(v) for v in os.environ.get("VARIANTS", "0,1,4,6").split(",")]
    torch.cuda.init()
    lib = _lib.get_dev_lib()
    us = C.c_double()
    if what in ("gemm", "gemm_l4", "gemm_mid"):
        for name, M, N, K, epi in {"gemm": GEMM_SHAPES, "gemm_l4": L4_SHAPES, "gemm_mid": MID_SHAPES}[what]:
            for mode in ("fast", "parity"):
                row = []
                for v in variants:
                    _lib.check(lib.cwm_debug_set(b"gemm_tile", v), lib)
                    best = 1e30
                    for _ in range(3):
                        _lib.check(lib.cwm_bench_gemm(M, N, K, _lib.mode_id(mode), epi, 20, C.byref(us)), lib)
                        best = min(best, us.value)
                    row.append("t%d %7.1f us %6.1f TF" % (v, best, 2.0 * M * N * K / best / 1e6))
                print("%-12s %-6s M=%d N=%d K=%d  %s" % (name, mode, M, N, K, " | ".join(row)), flush=True)
    else:
        shapes = ATTN_SHAPES
        if what == "attn_sweep":  # ~1.6e9 score elements per launch at every length (B*H*N*N), like the B/8 encoder launch of the bench batch
            shapes = [("n%d" % n, max(1, min(4096, int(round(2.4e8 / (n * n) / 12 * 12)))), 12, n) for n in (40, 64, 100, 128, 196, 256, 392, 512, 792, 1024, 1568, 2048)]
            shapes = [(nm, max(1, b // 12), 12, n) for nm, b, _, n in shapes]
        for name, B, H, N in shapes:
            for mode in ("fast", "parity"):
                row = []
                for kern in [int(v) for v in os.environ.get("ATTN_KERNELS", "1,3").split(",")]:
                    _lib.check(lib.cwm_debug_set(b"attn_kernel", kern), lib)
                    best = 1e30
                    for _ in range(3):
                        _lib.check(lib.cwm_bench_attention(B, H, N, _lib.mode_id(mode), 10, C.byref(us)), lib)
                        best = min(best, us.value)
                    row.append("k%d %8.1f us %7.1f TF" % (kern, best, 4.0 * B * H * N * N * 64 / best / 1e6))
                print("%-8s %-6s B=%d H=%d N=%d  %s" % (name, mode, B, H, N, " | ".join(row)), flush=True)
        _lib.check(lib.cwm_debug_set(b"attn_kernel", 0), lib)


if __name__ == "__main__":
    main()
