"""Small-batch latency A/B of one development switch: python tools/latency_ab.py KEY VA VB   (B = 1, 2, 4, 8; parity mode)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import _lib, config as C, synthetic as S, vmae
key, va, vb = sys.argv[1].encode(), int(sys.argv[2]), int(sys.argv[3])
cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
m = vmae.PretrainVisionTransformer(cfg, mode="parity")
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
lib = _lib.get_dev_lib()
for B in (1, 2, 4, 8):
    x = torch.from_numpy(S.synthetic_frames(B, cfg, 0)).cuda()
    mask = torch.from_numpy(S.synthetic_masks(B, cfg, 8, 0)).cuda()
    for rep in range(2):
        for v in (va, vb):
            m.set_option(key.decode() if isinstance(key, bytes) else key, v)
            for _ in range(5): m.predict_video(x, mask, n_vis=792, check=False)
            torch.cuda.synchronize(); n = 40; t0 = time.perf_counter()
            for _ in range(n): m.predict_video(x, mask, n_vis=792, check=False)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            print("B=%d %s=%d  %.3f ms / forward" % (B, key.decode(), v, 1e3 * dt), flush=True)
