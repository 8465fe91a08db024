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
