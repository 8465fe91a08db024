"""Step time over batch sizes (ViT-B/8 or CFG=..., parity, library defaults):  python tools/batch_sweep.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import config as C, synthetic as S, vmae
cfg = C.CONFIGS[os.environ.get("CFG", "base_8x8patch_2frames_1tube")]
kv, clump = (8, 1) if "base" in cfg.name else (32, 2)
m = vmae.PretrainVisionTransformer(cfg, mode=os.environ.get("MODE", "parity"))
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
nv = cfg.tokens_per_frame + kv
for B in [int(a) for a in sys.argv[1:]] or [8, 12, 16, 20, 24, 28, 32]:
    x = torch.from_numpy(S.synthetic_frames(B, cfg, 0)).cuda()
    mask = torch.from_numpy(S.synthetic_masks(B, cfg, kv, 0, clump)).cuda()
    best = 1e9
    for rep in range(3):
        for _ in range(3): m.predict_video(x, mask, n_vis=nv, check=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m.predict_video(x, mask, n_vis=nv, check=False)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20)
    print("B=%2d  %.3f ms  %.1f frames/s" % (B, 1e3 * best, B / best), flush=True)
