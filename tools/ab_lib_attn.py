"""Same-box A/B of two builds of the library on the attention shapes: python tools/ab_lib_attn.py LIB_A LIB_B  (each measured in its own child process,
alternating A B A B; a box's clock drifts less than the boxes differ)."""
import os, subprocess, sys
CHILD = r'''
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from counterfactualworldmodels_amd import _lib
from tools.microbench import ATTN_SHAPES
torch.cuda.init()
lib = _lib.get_lib(); us = C.c_double()
for name, B, H, N in ATTN_SHAPES + [("b8.enc.half", 16, 12, 792), ("b8.dec.half", 16, 6, 1568)]:
    for mode in ("parity", "fast"):
        best = 1e30
        for _ in range(3):
            _lib.check(lib.cwm_bench_attention(B, H, N, _lib.mode_id(mode), 30, C.byref(us)))
            best = min(best, us.value)
        print("%-12s %-6s %8.1f us %6.1f TF" % (name, mode, best, 4.0 * N * N * 64 * B * H / best / 1e6), flush=True)
'''
for rep in range(2):
    for tag, lib in (("A", sys.argv[1]), ("B", sys.argv[2])):
        env = dict(os.environ, CWM_HIP_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        for line in out.stdout.strip().split("\n"):
            print("%s%d %s %s" % (tag, rep, os.path.basename(lib), line), flush=True)
        if out.returncode:
            print(out.stderr[-2000:])
