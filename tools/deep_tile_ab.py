"""64x128 against 128x128 tiles of the deep-ring GEMM kernel over the small batches ("gemm_debug" bit 8 keeps 128 rows):  python tools/deep_tile_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import _lib, config as C, synthetic as S, vmae
cfg = C.CONFIGS[os.environ.get("CFG", "base_8x8patch_2frames_1tube")]
kv, clump = (8, 1) if "base" in cfg.name else (32, 2)
m = vmae.PretrainVisionTransformer(cfg, mode=os.environ.get("MODE", "parity"))
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
lib = _lib.get_lib()
nv = cfg.tokens_per_frame + kv
for B in [int(b) for b in os.environ.get("BATCHES", "1,2,3,4,6,8").split(",")]:
    x = torch.from_numpy(S.synthetic_frames(B, cfg, 0)).cuda()
    mask = torch.from_numpy(S.synthetic_masks(B, cfg, kv, 0, clump)).cuda()
    res = {}
    for rep in range(3):
        for bits in (256, 0):
            _lib.check(lib.cwm_debug_set(b"gemm_debug", bits))
            for _ in range(3): m.predict_video(x, mask, n_vis=nv, check=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): m.predict_video(x, mask, n_vis=nv, check=False)
            torch.cuda.synchronize(); res[bits] = min(res.get(bits, 1e9), (time.perf_counter() - t0) / 30)
    print("B=%2d  128-row tiles %.3f ms   64-row where half the CUs would idle %.3f ms   ratio %.3f" % (B, 1e3 * res[256], 1e3 * res[0], res[0] / res[256]), flush=True)
_lib.check(lib.cwm_debug_set(b"gemm_debug", 0))
