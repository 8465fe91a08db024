import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from counterfactualworldmodels_amd import config as C, synthetic as S, vmae, segmentation
cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
m = vmae.PretrainVisionTransformer(cfg, mode="parity")
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
B=32
x = torch.from_numpy(S.synthetic_frames(B, cfg, 0)).cuda()
mask = torch.from_numpy(S.synthetic_masks(B, cfg, 8, 0, 1)).cuda()
nv = cfg.tokens_per_frame + 8
G = segmentation.FlowGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
m.predict_video(x, mask, n_vis=nv)
def t(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
for rep in range(3):
    print("wrapper predict      %.3f ms" % t(lambda: G.predict(x, mask, frame=None)))
    print("predict_video nochk  %.3f ms" % t(lambda: m.predict_video(x, mask, n_vis=nv, check=False)))
    print("predict_video check  %.3f ms" % t(lambda: m.predict_video(x, mask, n_vis=nv, check=True)))
