"""Step time against the number of batch lanes (cwm_model_set_lanes 1 .. 4):  python tools/lanes_ab.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import config as C, synthetic as S, vmae
cfg = C.CONFIGS[os.environ.get("CFG", "base_8x8patch_2frames_1tube")]
kv, clump = (8, 1) if "base" in cfg.name else (32, 2)
B = int(sys.argv[1]) if len(sys.argv) > 1 else (32 if "base" in cfg.name else 8)
m = vmae.PretrainVisionTransformer(cfg, mode="parity")
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
x = torch.from_numpy(S.synthetic_frames(B, cfg, 0)).cuda()
mask = torch.from_numpy(S.synthetic_masks(B, cfg, kv, 0, clump)).cuda()
nv = cfg.tokens_per_frame + kv
m.predict_video(x, mask, n_vis=nv)
for rep in range(3):
    for lanes in (1, 2, 3, 4):
        m.set_lanes(lanes)
        for _ in range(4): m.predict_video(x, mask, n_vis=nv, check=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m.predict_video(x, mask, n_vis=nv, check=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print("B=%d lanes %d: %.3f ms  %.0f frames/s" % (B, lanes, 1e3 * dt, B / dt), flush=True)
