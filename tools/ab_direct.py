"""Same-box A/B of the GEMM epilogue forms: LDS-staged vs direct (`gemm_direct`), bitwise equality first, then timing per model shape.

    python tools/ab_direct.py            # B/8 shapes, parity mode;  SHAPES=l4|mid|b1  MODES=parity,fast  TILES=0,1,4
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

from counterfactualworldmodels_amd import _lib  # noqa: E402
from tools.microbench import GEMM_SHAPES, MID_SHAPES, L4_SHAPES, B1_SHAPES  # noqa: E402

torch.cuda.init()
lib = _lib.get_dev_lib()


def setk(key, v):
    _lib.check(lib.cwm_debug_set(key, v))


def check_bitwise():
    import gpu_utils as gu

    def rnd(*shape, seed=0, scale=1.0):
        g = torch.Generator().manual_seed(seed)
        return torch.randn(*shape, generator=g) * scale

    bad = 0
    for tile in (1, 4, 6):
        setk(b"gemm_tile", tile)
        for mode in ("parity", "fast"):
            for (M, N, K) in [(300, 272, 128), (1000, 1152, 192), (77, 48, 512), (513, 400, 384), (2049, 768, 768)]:
                a, w, b, r = rnd(M, K, seed=21), rnd(N, K, seed=22, scale=K ** -0.5), rnd(N, seed=23), rnd(M, N, seed=24)
                outs = []
                for direct in (0, 1):
                    setk(b"gemm_direct", direct)
                    outs.append((gu.linear(a, w, b, mode=mode), gu.linear(a, w, b, resid=r, mode=mode), gu.linear(a, w, b, gelu=True, mode=mode),
                                 gu.linear(a, w, None, mode=mode)))
                for idx, (o0, o1) in enumerate(zip(*outs)):
                    if not torch.equal(o0, o1):
                        bad += 1
                        print("MISMATCH tile %d %s M=%d N=%d K=%d out %d maxdiff %g" % (tile, mode, M, N, K, idx, (o0 - o1).abs().max().item()), flush=True)
    setk(b"gemm_tile", 0)
    setk(b"gemm_direct", 0)
    print("bitwise check: %s" % ("OK" if bad == 0 else "%d mismatches" % bad), flush=True)
    return bad


def timing():
    us = C.c_double()
    shapes = {"b8": GEMM_SHAPES, "mid": MID_SHAPES, "l4": L4_SHAPES, "b1": B1_SHAPES}[os.environ.get("SHAPES", "b8")]
    modes = os.environ.get("MODES", "parity").split(",")
    tiles = [int(t) for t in os.environ.get("TILES", "0,1,4").split(",")]
    for name, M, N, K, epi in shapes:
        for mode in modes:
            cells = []
            for tile in tiles:
                setk(b"gemm_tile", tile)
                pair = []
                for direct in (0, 1):
                    setk(b"gemm_direct", direct)
                    best = 1e30
                    for _ in range(3):
                        _lib.check(lib.cwm_bench_gemm(M, N, K, _lib.mode_id(mode), epi, 20, C.byref(us)))
                        best = min(best, us.value)
                    pair.append(best)
                cells.append("t%d staged %6.1f direct %6.1f us (%+5.1f %%) %5.0f TF" % (tile, pair[0], pair[1], 100.0 * (pair[1] / pair[0] - 1.0),
                                                                                       2.0 * M * N * K / pair[1] / 1e6))
            print("%-14s %-6s %s" % (name, mode, " | ".join(cells)), flush=True)
    setk(b"gemm_tile", 0)
    setk(b"gemm_direct", 0)


if __name__ == "__main__":
    if "--no-check" not in sys.argv:
        check_bitwise()
    timing()
