"""Experiment (GPU box): two half-batches on two HIP streams vs one full batch (does cross-stream overlap fill the kernel tails?)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from counterfactualworldmodels_amd import config as C, synthetic as S, vmae  # noqa: E402

cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
W = {k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()}


def model():
    m = vmae.PretrainVisionTransformer(cfg, mode="parity")
    m.load_state_dict(W)
    return m.cuda().eval()


B = 32
x = torch.from_numpy(S.synthetic_frames(B, cfg, 0)).cuda()
mask = torch.from_numpy(S.synthetic_masks(B, cfg, 8, 0)).cuda()
m0, m1, m2, m3 = model(), model(), model(), model()
m3.predict_video(x, mask, n_vis=792, check=False)
m3.set_lanes(1)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h = B // 2


def full(n):
    for _ in range(n):
        m0.predict_video(x, mask, n_vis=792, check=False)


def full_one_lane(n):
    for _ in range(n):
        m3.predict_video(x, mask, n_vis=792, check=False)


def two_full(n):
    for _ in range(n):
        with torch.cuda.stream(s1):
            m1.predict_video(x, mask, n_vis=792, check=False)
        with torch.cuda.stream(s2):
            m2.predict_video(x, mask, n_vis=792, check=False)


def halves(n):
    for _ in range(n):
        with torch.cuda.stream(s1):
            m1.predict_video(x[:h], mask[:h], n_vis=792, check=False)
        with torch.cuda.stream(s2):
            m2.predict_video(x[h:], mask[h:], n_vis=792, check=False)


for name, fn, nb in (("library lanes=2, batch 32", full, 1), ("library lanes=1, batch 32", full_one_lane, 1), ("two streams, batch 16 each", halves, 1), ("two streams, batch 32 each", two_full, 2), ("library lanes=2, batch 32", full, 1)):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(15)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-28s %.2f ms / step  %.1f frames/s" % (name, 1e3 * dt / 15, nb * B * 15 / dt), flush=True)
