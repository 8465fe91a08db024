"""Per-phase s_memtime totals of the software-pipelined attention kernel (block 0, wave 0); needs a library built with
CWM_HIPCC_EXTRA=-DCWM_ATTN_PROF.   python tools/attn_prof.py B H N mode"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from counterfactualworldmodels_amd import _lib  # noqa: E402

torch.cuda.init()
lib = _lib.get_dev_lib()
us = C.c_double()
B, H, N = (int(v) for v in sys.argv[1:4])
_lib.check(lib.cwm_debug_set(b"attn_kernel", 3))
_lib.check(lib.cwm_bench_attention(B, H, N, _lib.mode_id(sys.argv[4]), 3, C.byref(us)))
t = [lib.cwm_debug_set(b"attn_prof", i) for i in range(4)]
tot, real = lib.cwm_debug_set(b"attn_prof", 4), lib.cwm_debug_set(b"attn_prof", 5)
print("block 0: %d s_memtime ticks in %d s_memrealtime ticks (100 MHz) -> %.0f ticks/us" % (tot, real, tot / (real / 100.0)))
nkt = (N + 63) // 64
print(sys.argv[1:], "%.1f us; ticks per tile: stage %.0f  phase A %.0f  phase B %.0f  wait+barrier %.0f  (sum %.0f)" % (
    us.value, t[0] / nkt, t[1] / nkt, t[2] / nkt, t[3] / nkt, sum(t) / nkt))

import numpy as np
if lib.cwm_debug_set(b"attn_prof", 1000) == 0:
    nblk = B * H * ((N + 127) // 128)
    r = np.fromfile("/tmp/attn_blocks.bin", dtype=np.uint64).reshape(-1, 8)[:nblk].astype(np.int64)
    n_all = len(r)
    xcd_of = (np.arange(n_all) & 7)[r[:, 0] > 0]  # workgroups are dealt round-robin over the XCDs by linear id
    r = r[r[:, 0] > 0]  # workgroups of the ragged query tile (attention_tail.h) leave no record
    nblk = len(r)
    print("workgroups with a record: %d of %d" % (nblk, n_all))
    t0 = r[:, 0].min()
    start, end, cyc, hw = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0, r[:, 2], r[:, 3]
    dur = end - start
    print("blocks %d: kernel span %.1f us; block duration us: min %.1f  median %.1f  max %.1f; cycles median %.0f" % (nblk, end.max(), dur.min(), np.median(dur), dur.max(), np.median(cyc)))
    order = np.argsort(start)
    for lo in range(0, nblk, max(1, nblk // 12)):
        sel = order[lo:lo + max(1, nblk // 12)]
        print("  blocks started %7.1f..%7.1f us: n %4d  duration %.1f us  clock %.2f GHz" % (start[sel].min(), start[sel].max(), len(sel), dur[sel].mean(), (cyc[sel] / dur[sel]).mean() / 1e3))
    # concurrency: number of blocks alive over time
    ts = np.linspace(0, end.max(), 13)[1:-1]
    print("  alive blocks at", " ".join("%.0fus:%d" % (t, ((start <= t) & (end > t)).sum()) for t in ts))
    print("  sum of workgroup durations / 512 slots: %.1f us; last start %.1f us; ends: p50 %.1f  p90 %.1f  p99 %.1f  max %.1f us" % (
        dur.sum() / 512, start.max(), *np.percentile(end, [50, 90, 99]), end.max()))
    for x in range(8):
        sel = xcd_of == x
        st, en, du = start[sel], end[sel], dur[sel]
        o = np.argsort(st)
        print("  XCD %d: n %4d  mean duration %.1f us  clock %.2f GHz  first-64 mean %.1f  last end %.1f us  busy slot-time %.1f us of 64 slots  starts #64/#128/#192/#256: %s" % (
            x, sel.sum(), du.mean(), (cyc[sel] / du).mean() / 1e3, du[o[:64]].mean(), en.max(), du.sum() / 64,
            " ".join("%.1f" % st[o[k]] for k in (64, 128, 192, 256) if k < len(o))))
    print("  distinct (xcc, se, cu) ids:", len(set((int(h) >> 8) & 0xfffff for h in hw)))
    ph = r[:, 4:8] / float(nkt)
    print("  per-workgroup phase cycles per tile (wave 0): median stage/DMA %.0f  phase A %.0f  phase B %.0f  wait+barrier %.0f   (p90: %.0f %.0f %.0f %.0f)" % (
        *np.median(ph, 0), *np.percentile(ph, 90, 0)))
