"""Step-level tuning of the GEMM tile choice (development tool): coordinate descent over the tile configuration of every GEMM shape of a
forward, timing the whole step (two batch lanes by default), through the cwm_gemm_tile_override hook.

    python tools/autotune_step.py            # ViT-B/8 batch 32, two lanes, parity;  CFG=large_4x4patch_2frames_1tube LANES=1 ...

Prints the shapes, the default choice's step time, every trial, and the overrides that beat the default by more than the noise.  All
configurations give bit-identical results, so whatever it finds is speed only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from counterfactualworldmodels_amd import _lib, config as C, synthetic as S, vmae  # noqa: E402

lanes = int(os.environ.get("LANES", "2"))
cfg = C.CONFIGS[os.environ.get("CFG", "base_8x8patch_2frames_1tube")]
B, kv, clump = (32, 8, 1) if "base" in cfg.name else (8, 32, 2)
B = int(os.environ.get("BATCH", B))
m = vmae.PretrainVisionTransformer(cfg, mode=os.environ.get("MODE", "parity"))
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
x = torch.from_numpy(S.synthetic_frames(B, cfg, 0)).cuda()
mask = torch.from_numpy(S.synthetic_masks(B, cfg, kv, 0, clump)).cuda()
nv = cfg.tokens_per_frame + kv
lib = _lib.get_dev_lib()
_lib.check(lib.cwm_gemm_tile_override(0, 0, 0, 0, 0, 0), lib)  # installs the per-shape hook in this thread's options ...
m.use_library(lib)                                             # ... which the model's handle, created in the dev library, starts from
m.predict_video(x, mask, n_vis=nv)
m.set_lanes(lanes)


def run(steps=20):
    for _ in range(4):
        m.predict_video(x, mask, n_vis=nv, check=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.predict_video(x, mask, n_vis=nv, check=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


# the GEMM shapes of one lane (engine.hip run_block / model.hip forward_lane): (name, M, N, K, epi)
two = lanes >= 2 and (B // 2) * nv >= 3000  # engine.h kMinLaneRows
Bl = (B + 1) // 2 if two else B
De, Dd, Nt, Nm = cfg.enc_dim, cfg.dec_dim, cfg.num_tokens, cfg.num_tokens - nv
r64 = lambda v: (v + 63) // 64 * 64
shapes = [("enc.qkv", Bl * nv, 3 * De, De, 3), ("enc.proj", Bl * nv, De, De, 0), ("enc.fc1", Bl * nv, 4 * De, De, 1), ("enc.fc2", Bl * nv, De, 4 * De, 0),
          ("dec.qkv", Bl * Nt, 3 * Dd, Dd, 3), ("dec.proj", Bl * Nt, Dd, Dd, 0), ("dec.fc1", Bl * Nt, 4 * Dd, Dd, 1), ("dec.fc2", Bl * Nt, Dd, 4 * Dd, 0),
          ("dec.last.proj", Bl * Nm, Dd, Dd, 0), ("dec.last.fc1", Bl * Nm, 4 * Dd, Dd, 1), ("dec.last.fc2", Bl * Nm, Dd, 4 * Dd, 0),
          ("e2d", Bl * nv, Dd, De, 0)]
if two and B % 2:
    print("(odd batch: the second lane's shapes differ by one sample and keep the default rule)")
ovl = 1 if two else 0
base = min(run(), run())
print("%s batch %d lanes %d (%s): default tile rule %.3f ms/step" % (cfg.name, B, lanes, "two-lane call" if two else "one lane", base), flush=True)
best = {}
cur = base
for rnd in range(2):
    improved = False
    for name, M, N, K, epi in shapes:
        keep = best.get(name, 0)
        trials = {}
        for cand in (1, 4, 6):
            if cand == keep:
                continue
            _lib.check(lib.cwm_gemm_tile_override(M, N, r64(K), epi, ovl, cand))
            trials[cand] = min(run(), run())
            _lib.check(lib.cwm_gemm_tile_override(M, N, r64(K), epi, ovl, keep))
        c, t = min(trials.items(), key=lambda kv_: kv_[1])
        print("  round %d %-14s M=%6d N=%5d K=%5d  %s   (current %.3f)" % (rnd, name, M, N, K, "  ".join("cfg %d: %.3f" % kv_ for kv_ in sorted(trials.items())), cur),
              flush=True)
        if t < cur * 0.997:  # beyond the run-to-run noise of a 20-step average
            best[name] = c
            cur = t
            improved = True
            _lib.check(lib.cwm_gemm_tile_override(M, N, r64(K), epi, ovl, c))
    if not improved:
        break
final = min(run(), run(), run())
_lib.check(lib.cwm_gemm_tile_override(0, 0, 0, 0, 0, 0))
again = min(run(), run(), run())
print("overrides kept: %s" % ({k: v for k, v in best.items()} or "none"))
print("step with the overrides %.3f ms  |  default rule (re-measured) %.3f ms  ->  %+.2f %%" % (final, again, 100.0 * (again / final - 1.0)))
