"""Same-box A/B of the step over chain configurations of the pipelined attention kernel:  python tools/ab_chain_step.py [lanes] [C:heads ...]
(heads: -1 all, -2 whole rounds only; alternates the configurations three times)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import _lib, config as C, synthetic as S, vmae
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfgs = [tuple(int(v) for v in a.split(":")) for a in sys.argv[2:]] or [(1, -1), (3, -1), (2, -1), (2, -2), (3, -2)]
cfg = C.CONFIGS[os.environ.get("CFG", "base_8x8patch_2frames_1tube")]
B, kv, clump = (32, 8, 1) if "base" in cfg.name else (8, 32, 2)
B = int(os.environ.get("BATCH", B))
m = vmae.PretrainVisionTransformer(cfg, mode=os.environ.get("MODE", "parity"))
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
x = torch.from_numpy(S.synthetic_frames(B, cfg, 0)).cuda()
mask = torch.from_numpy(S.synthetic_masks(B, cfg, kv, 0, clump)).cuda()
nv = cfg.tokens_per_frame + kv
lib = _lib.get_lib()
ref = m.predict_video(x, mask, n_vis=nv)[1].clone()
m.set_lanes(lanes)
def run():
    for _ in range(5): m.predict_video(x, mask, n_vis=nv, check=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m.predict_video(x, mask, n_vis=nv, check=False)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 20
res = {}
for rep in range(3):
    for (cl, hc) in cfgs:
        _lib.check(lib.cwm_debug_set(b"attn_chain", cl)); _lib.check(lib.cwm_debug_set(b"attn_chain_heads", hc))
        same = torch.equal(m.predict_video(x, mask, n_vis=nv)[1], ref)
        dt = run()
        res.setdefault((cl, hc), []).append(dt)
        print("%s batch %d lanes %d chain %d heads %d: %.3f ms/step  %.0f frames/s  bitwise %s" % (cfg.name[:8], B, lanes, cl, hc, 1e3 * dt, B / dt, same), flush=True)
base = min(res[cfgs[0]])
for k, v in res.items():
    print("chain %d heads %d: best %.3f ms  (%+.2f %%)" % (k[0], k[1], 1e3 * min(v), 100 * (min(v) / base - 1)))
