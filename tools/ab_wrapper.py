"""Step time through the wrapper's predict (rectangulariser sync per call) vs the bare library call, same box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import config as C, segmentation, synthetic as S, vmae
cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
m = vmae.PretrainVisionTransformer(cfg, mode="parity")
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
G = segmentation.FlowGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
x = torch.from_numpy(S.synthetic_frames(32, cfg, 0)).cuda()
mask = torch.from_numpy(S.synthetic_masks(32, cfg, 8, 0)).cuda()
m.predict_video(x, mask, n_vis=792)
def run(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 20
for rep in range(3):
    a = run(lambda: G.predict(x, mask, frame=None))
    b = run(lambda: m.predict_video(x, mask, n_vis=792, check=False))
    print("wrapper predict %.3f ms (%.0f frames/s) | bare library call %.3f ms (%.0f frames/s)" % (1e3 * a, 32 / a, 1e3 * b, 32 / b), flush=True)
