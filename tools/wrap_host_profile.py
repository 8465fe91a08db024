"""Host-side cost of one wrapper call (python + ctypes + library host code), timed per segment with the GPU queue empty: where do the microseconds between
the mask read-back and the first kernel launch go?  (tools/wrap_gap.py measures the resulting idle gap on the GPU from a kernel trace.)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from counterfactualworldmodels_amd import _lib, config as C, synthetic as S, vmae, segmentation, masking
cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
m = vmae.PretrainVisionTransformer(cfg, mode="parity")
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
x = torch.from_numpy(S.synthetic_frames(32, cfg, 0)).cuda()
mask = torch.from_numpy(S.synthetic_masks(32, cfg, 8, 0, 1)).cuda()
G = segmentation.FlowGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
for _ in range(3): G.predict(x, mask, frame=None)
torch.cuda.synchronize()
lib = _lib.get_lib()
real_forward = lib.cwm_forward
stamps = {}
def fwd(*a):
    stamps["enter"] = time.perf_counter()
    rc = real_forward(*a)
    stamps["leave"] = time.perf_counter()
    return rc
rect = G.mask_rectangularizer
real_to_host = rect._to_host
def to_host(mk):
    stamps["copy0"] = time.perf_counter()
    h = real_to_host(mk)
    stamps["copy1"] = time.perf_counter()
    return h
rect._to_host = to_host
class L:  # proxy: everything but cwm_forward straight from the library
    def __getattr__(self, k): return fwd if k == "cwm_forward" else getattr(lib, k)
_lib.get_lib = lambda: L()
acc = {}
N = 40
for _ in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    G.predict(x, mask, frame=None)
    t1 = time.perf_counter()
    for k, v in (("before the read-back (sync_weights, output allocation)", stamps["copy0"] - t0), ("read-back: copy + wait", stamps["copy1"] - stamps["copy0"]),
                 ("read-back -> cwm_forward entry (counts, python, ctypes)", stamps["enter"] - stamps["copy1"]), ("cwm_forward (host enqueue of the forward)", stamps["leave"] - stamps["enter"]),
                 ("after cwm_forward", t1 - stamps["leave"])):
        acc[k] = acc.get(k, 0.0) + v
for k, v in acc.items():
    print("%-62s %7.1f us" % (k, 1e6 * v / N))
