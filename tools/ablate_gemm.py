import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import _lib
torch.cuda.init(); lib = _lib.get_lib(); us = C.c_double()
for (name, M, N, K, epi) in [("fc1", 25344, 3072, 768, 1), ("fc2", 25344, 768, 3072, 0), ("fc2big", 25344, 3072, 3072, 0)]:
    for mode in ("fast", "parity"):
        row = []
        for ab in (0, 1, 2, 4, 5, 3):
            _lib.check(lib.cwm_debug_set(b"gemm_ablate", ab))
            best = 1e30
            for _ in range(3):
                _lib.check(lib.cwm_bench_gemm(M, N, K, _lib.mode_id(mode), epi, 10, C.byref(us))); best = min(best, us.value)
            row.append("ab%d %7.1f" % (ab, best))
        print(name, mode, " | ".join(row), flush=True)
_lib.check(lib.cwm_debug_set(b"gemm_ablate", 0))
