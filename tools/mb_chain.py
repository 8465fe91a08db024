"""Chained work items of the pipelined attention kernel ("attn_chain" = C query tiles per workgroup, "attn_chain_heads" = chained heads per XCD):
bitwise check against single-tile items, then a timing sweep on the model shapes (same box, alternating).

    python tools/mb_chain.py [check] [sweep]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from counterfactualworldmodels_amd import _lib  # noqa: E402

torch.cuda.init()
lib = _lib.get_lib()
dbg = getattr(lib, "cwm_debug_set")


def setk(key, v):
    _lib.check(dbg(key.encode(), v))


def attention(qkv_d, H, mode):
    B, N, _ = qkv_d.shape
    out = torch.empty(B, N, H * 64, device="cuda", dtype=torch.float32)
    _lib.check(lib.cwm_attention(qkv_d.data_ptr(), out.data_ptr(), B, N, H, _lib.mode_id(mode), _lib.current_stream_handle(torch.device("cuda:0"))))
    return out


def check():
    bad = 0
    setk("attn_kernel", 3)
    for (B, H, N) in [(2, 12, 792), (4, 6, 1568), (1, 16, 3168), (1, 8, 1024), (2, 4, 640), (1, 8, 256), (3, 8, 520), (2, 8, 6272)]:
        g = torch.Generator().manual_seed(N)
        qkv = (torch.randn(B, N, 3 * H * 64, generator=g) * 1.5).cuda()
        for mode in ("parity", "fast"):
            setk("attn_chain", 1)
            ref = attention(qkv, H, mode)
            for Cn in (2, 3, 4, 6, 7):
                for hc in (-1, 1):
                    setk("attn_chain", Cn)
                    setk("attn_chain_heads", hc)
                    for rep in range(3):
                        out = attention(qkv, H, mode)
                        same = torch.equal(out, ref)
                        if not same:
                            bad += 1
                            print("MISMATCH B=%d H=%d N=%d %s chain=%d heads=%d rep=%d maxdiff=%g" % (B, H, N, mode, Cn, hc, rep, (out - ref).abs().max().item()), flush=True)
                            break
        print("checked B=%d H=%d N=%d  (bad so far: %d)" % (B, H, N, bad), flush=True)
    setk("attn_chain", 0)
    setk("attn_chain_heads", -1)
    setk("attn_kernel", 0)
    return bad


def sweep():
    us = C.c_double()
    shapes = [("b8.enc", 32, 12, 792), ("b8.enc.half", 16, 12, 792), ("b8.dec", 32, 6, 1568), ("b8.dec.half", 16, 6, 1568), ("l4.enc", 8, 16, 3168), ("l4.enc.half", 4, 16, 3168)]
    for name, B, H, N in shapes:
        per = B * H // 8
        full = N // 128
        cfgs = [(1, -1)]
        for Cn in (2, 3, 4, 6, 12):
            if Cn > full:
                continue
            nc = full // Cn
            # chained heads per XCD: all, or as many as fill whole rounds of the 64 slots of an XCD
            hcs = {per}
            for rounds in (1, 2, 3):
                hc = (64 * rounds) // nc
                if 0 < hc < per:
                    hcs.add(hc)
            for hc in sorted(hcs):
                cfgs.append((Cn, hc))
        for mode in ("parity",):
            res = {}
            for rep in range(2):
                for (Cn, hc) in cfgs:
                    setk("attn_chain", Cn)
                    setk("attn_chain_heads", hc)
                    _lib.check(lib.cwm_bench_attention(B, H, N, _lib.mode_id(mode), 30, C.byref(us)))
                    res.setdefault((Cn, hc), []).append(us.value)
            base = min(res[(1, -1)])
            print("%-12s %-6s B=%d H=%d N=%d (heads/XCD %d, full tiles %d)" % (name, mode, B, H, N, per, full), flush=True)
            for k, v in res.items():
                print("    chain %2d heads %3d : %s  best %7.1f us  (%+.1f %%)" % (k[0], k[1], " ".join("%7.1f" % x for x in v), min(v), 100.0 * (min(v) / base - 1)), flush=True)
    setk("attn_chain", 0)
    setk("attn_chain_heads", -1)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "sweep"]
    rc = 0
    if "check" in what:
        rc = 1 if check() else 0
    if "sweep" in what:
        sweep()
    sys.exit(rc)
