"""B = 1 forward in a loop (for rocprofv3 --kernel-trace --stats): python tools/latency_b1.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from counterfactualworldmodels_amd import config as C, synthetic as S, vmae  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
m = vmae.PretrainVisionTransformer(cfg, mode="parity")
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
x = torch.from_numpy(S.synthetic_frames(B, cfg, 0)).cuda()
mask = torch.from_numpy(S.synthetic_masks(B, cfg, 8, 0)).cuda()
for _ in range(20):
    m.predict_video(x, mask, n_vis=792, check=False)
torch.cuda.synchronize()
