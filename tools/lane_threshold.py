"""Where does the second batch lane start to pay?  python tools/lane_threshold.py  (ViT-B/8, parity; cwm_debug_set "min_lane_rows")"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import _lib, config as C, synthetic as S, vmae
cfg = C.CONFIGS[os.environ.get("CFG", "base_8x8patch_2frames_1tube")]
kv, clump = (8, 1) if "base" in cfg.name else (32, 2)
m = vmae.PretrainVisionTransformer(cfg, mode="parity")
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
lib = _lib.get_dev_lib()
nv = cfg.tokens_per_frame + kv
for B in [int(b) for b in os.environ.get("BATCHES", "4,6,8,10,12,14,16").split(",")]:
    x = torch.from_numpy(S.synthetic_frames(B, cfg, 0)).cuda()
    mask = torch.from_numpy(S.synthetic_masks(B, cfg, kv, 0, clump)).cuda()
    res = []
    for rows in (1 << 30, 1):
        m.set_option("min_lane_rows", rows)
        best = 1e9
        for rep in range(3):
            for _ in range(3): m.predict_video(x, mask, n_vis=nv, check=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): m.predict_video(x, mask, n_vis=nv, check=False)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20)
        res.append(best)
    print("B=%2d (%5d encoder rows per half)  one lane %.3f ms   two lanes %.3f ms   ratio %.3f" % (B, (B // 2) * nv, 1e3 * res[0], 1e3 * res[1], res[1] / res[0]), flush=True)
m.set_option("min_lane_rows", 3000)  # engine.h kMinLaneRows
