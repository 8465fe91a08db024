"""Same-box A/B of the B/8 batch-32 step for one development switch:  python tools/ab_step.py KEY VALUE_A VALUE_B [lanes]
(alternates A B A B; results of different GPU boxes differ by +-3 %, so only same-process comparisons count)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import _lib, config as C, synthetic as S, vmae
key, va, vb = sys.argv[1].encode(), int(sys.argv[2]), int(sys.argv[3])
lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 2
cfg = C.CONFIGS[os.environ.get("CFG", "base_8x8patch_2frames_1tube")]
B, kv, clump = (32, 8, 1) if "base" in cfg.name else (8, 32, 2)
m = vmae.PretrainVisionTransformer(cfg, mode=os.environ.get("MODE", "parity"))
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
x = torch.from_numpy(S.synthetic_frames(B, cfg, 0)).cuda()
mask = torch.from_numpy(S.synthetic_masks(B, cfg, kv, 0, clump)).cuda()
nv = cfg.tokens_per_frame + kv
lib = _lib.get_dev_lib()
m.predict_video(x, mask, n_vis=nv)
m.set_lanes(lanes)
def run():
    for _ in range(5): m.predict_video(x, mask, n_vis=nv, check=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m.predict_video(x, mask, n_vis=nv, check=False)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 20
for rep in range(3):
    for v in (va, vb):
        m.set_option(key.decode() if isinstance(key, bytes) else key, v)
        dt = run()
        print("%s=%d lanes %d: %.3f ms/step  %.0f frames/s" % (key.decode(), v, lanes, 1e3 * dt, B / dt), flush=True)
