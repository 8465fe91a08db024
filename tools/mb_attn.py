"""A/B of an attention switch (default: the workgroup mapping, attn_remap 1 / 0; KEY=attn_tail for the key-split of the ragged last tile) on the model shapes, both modes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import _lib
from tools.microbench import ATTN_SHAPES
torch.cuda.init()
lib = _lib.get_dev_lib(); us = C.c_double()
for name, B, H, N in ATTN_SHAPES + [("b8.enc.half", 16, 12, 792), ("b8.dec.half", 16, 6, 1568), ("b8.dec.last", 32, 6, 1568)]:
    for mode in ("parity", "fast"):
        row = []
        for remap in (1, 0, 1, 0):  # value of the switch
            _lib.check(lib.cwm_debug_set(os.environ.get("KEY", "attn_remap").encode(), remap))
            _lib.check(lib.cwm_bench_attention(B, H, N, _lib.mode_id(mode), 30, C.byref(us)))
            row.append("on %d %7.1f us %6.1f TF" % (remap, us.value, 4.0 * N * N * 64 * B * H / us.value / 1e6))
        print("%-12s %-6s B=%d H=%d N=%d  %s" % (name, mode, B, H, N, " | ".join(row)), flush=True)
_lib.check(lib.cwm_debug_set(os.environ.get("KEY", "attn_remap").encode(), 1))
