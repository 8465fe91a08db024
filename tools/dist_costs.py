"""Per-rank fixed costs of the sharded 256-prompt loop, measured piece by piece on ONE GPU: what a rank does
besides its predictor calls.  (The collectives themselves need > 1 rank; their payloads are printed.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import config as C, dist as cdist, segmentation, synthetic as S, vmae
cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
dev = torch.device("cuda:0")
m = vmae.PretrainVisionTransformer(cfg, mode="parity")
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
G = segmentation.FlowGenerator(predictor=m.to(dev).eval(), imagenet_normalize_inputs=True, temporal_dim=2)
x0 = torch.from_numpy(S.synthetic_frames(1, cfg, 0))[:, 0:1].to(dev)
table = torch.from_numpy(S.synthetic_prompts(256, cfg, 0)).to(dev)
build, rect, predict = cdist.prompt_hooks(G, frame=-1)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out
t, (xa, ma) = timeit(lambda: build(x0, table)); print("one rank: build all 256 prompts (frames + masks)    %.3f ms" % t)
t, _ = timeit(lambda: build(x0, table, frames=False)); print("rank 0: build the MASKS of all 256 prompts (world > 1)   %.3f ms" % t)
t, (mr, nm) = timeit(lambda: rect(ma.clone())); print("rank 0: rectangularise 256 rows (one host read-back) %.3f ms" % t)
t, buf = timeit(lambda: cdist.pack_inputs(x0, table, mr, nm, dev)); print("rank 0: pack {header|frame|table|masks} = %d bytes  %.3f ms" % (buf.numel(), t))
t, _ = timeit(lambda: cdist.unpack_inputs(buf)); print("rank r: unpack (128-byte header read-back)            %.3f ms" % t)
t, (xs, _) = timeit(lambda: build(x0, table[:32])); print("rank r: build its 32 prompts                          %.3f ms" % t)
t, y = timeit(lambda: predict(xa[:32], mr[:32], nm, 32), 10); print("rank r: predict 32 prompts (one library call)         %.3f ms" % t)
print("all-gather payload per rank at 8 ranks: %.1f MB of %.1f MB" % (y.numel() * 4 / 1e6, 8 * y.numel() * 4 / 1e6))
t, _ = timeit(lambda: cdist.sharded_counterfactual_predictions(x0, table, build, rect, predict, dev, chunk=32, comm=cdist.LocalComm()), 5)
print("one rank, all 256 prompts end to end                     %.3f ms (%.0f prompts/s)" % (t, 256e3 / t))
# ---- latency view: rank 0's build path of a sharded call, piece by piece IN ORDER with the device idle before each piece (what the `build` span of
# per_rank_ms sees: host issue time + the read-back's round trip, not throughput)
def once(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, out
for rep in range(3):
    t_m, (_, m0) = once(lambda: build(x0, table, frames=False))
    t_r, (mr2, nm2) = once(lambda: rect(m0))
    t_p, _ = once(lambda: cdist.pack_inputs(x0, table, mr2, nm2, dev))
    t_o, _ = once(lambda: build(x0, table[:32]))
    print("latency view %d: masks %.3f  rectangularise %.3f  pack %.3f  (sum %.3f ms)  |  own 32 prompts %.3f" % (rep, t_m, t_r, t_p, t_m + t_r + t_p, t_o))
from counterfactualworldmodels_amd.masking import RectangularizeMasks
r = RectangularizeMasks("min")
t_c, counts = once(lambda: r._counts_to_host(m0))
t0 = time.perf_counter(); [torch.randperm(784) for _ in range(32)]; t_rp = (time.perf_counter() - t0) * 1e3
print("rectangulariser pieces: counts kernel + 1-KB read-back %.3f ms, 32 x torch.randperm(784) on the host %.3f ms" % (t_c, t_rp))
# ---- the `build` span of a whole call as bench.py reports it (per_rank_ms), and the host's own timeline through the same pieces without any synchronisation
for rep in range(3):
    ph = cdist.PhaseTimes(dev)
    torch.cuda.synchronize()
    cdist.sharded_counterfactual_predictions(x0, table, build, rect, predict, dev, chunk=32, comm=cdist.LocalComm(), times=ph)
    print("PhaseTimes of one call (one rank: frames of all 256 prompts in `build`):", {k: round(v, 3) for k, v in ph.result().items()})
torch.cuda.synchronize()
for rep in range(3):
    t = [time.perf_counter()]
    _, m1 = build(x0, table, frames=False); t.append(time.perf_counter())
    m1, n1 = rect(m1); t.append(time.perf_counter())
    b1 = cdist.pack_inputs(x0, table, m1, n1, dev); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    print("host timeline %d (ms since start): masks issued %.3f, rectangularised %.3f, packed %.3f, device done %.3f" % ((rep,) + tuple(1e3 * (v - t[0]) for v in t[1:])))
# ---- after heavy work: the same pieces right behind a 32-prompt predictor call (1100 launches), as they run in steady state; every piece timed on the host
import numpy as np
from counterfactualworldmodels_amd import _lib
xs32, _ = build(x0, table[:32])
def stamp(fn, acc, key):
    t0 = time.perf_counter(); out = fn(); acc[key] = acc.get(key, 0.0) + (time.perf_counter() - t0) * 1e3; return out
for heavy in (False, True, True):
    if heavy:
        predict(xs32, mr[:32], nm, 32)
    torch.cuda.synchronize()
    acc = {}
    _, m2 = stamp(lambda: build(x0, table, frames=False), acc, "masks (10 torch ops + 1 kernel)")
    r2 = RectangularizeMasks("min"); r2._stage, r2._counts_dev, r2._event = r._stage, r._counts_dev, r._event
    counts2 = stamp(lambda: r2._counts_to_host(m2), acc, "counts kernel + read-back + spin")
    stamp(lambda: [torch.randperm(784) for _ in range(32)], acc, "32 x randperm(784)")
    stamp(lambda: rect(m2), acc, "whole rectangulariser again (no row changes left: counts only)")
    stamp(lambda: cdist.pack_inputs(x0, table, m2, 783, dev), acc, "pack")
    stamp(lambda: torch.cuda.synchronize(), acc, "final sync")
    print("after %s work:" % ("a 32-prompt predictor call's" if heavy else "no"), {k: round(v, 3) for k, v in acc.items()})
