"""Small-batch latency of the fused predict path (development tool, GPU box): python tools/latency.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from counterfactualworldmodels_amd import config as C, synthetic as S, vmae  # noqa: E402

cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
m = vmae.PretrainVisionTransformer(cfg, mode="parity")
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
for B in (1, 2, 4, 8, 32):
    x = torch.from_numpy(S.synthetic_frames(B, cfg, 0)).cuda()
    mask = torch.from_numpy(S.synthetic_masks(B, cfg, 8, 0)).cuda()
    for _ in range(3):
        m.predict_video(x, mask, n_vis=792, check=False)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        m.predict_video(x, mask, n_vis=792, check=False)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("B=%2d  %.3f ms / forward  (host issue %.3f ms)  %.1f frames/s" % (B, 1e3 * dt / n, 1e3 * t_issue / n, B * n / dt), flush=True)
