"""Same-box comparison of several switch settings on the B/8 batch-32 step:  python tools/ab_multi.py LANES "k1=v1,k2=v2" "k1=v3" ...
(each setting = a comma list of cwm_debug_set key=value pairs applied on top of the defaults; settings alternate, 3 rounds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import _lib, config as C, synthetic as S, vmae
lanes = int(sys.argv[1])
settings = [[kv.split("=") for kv in a.split(",") if kv] for a in sys.argv[2:]]
keys = sorted({k for st in settings for k, _ in st})
cfg = C.CONFIGS[os.environ.get("CFG", "base_8x8patch_2frames_1tube")]
B, kv, clump = (32, 8, 1) if "base" in cfg.name else (8, 32, 2)
m = vmae.PretrainVisionTransformer(cfg, mode=os.environ.get("MODE", "parity"))
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
x = torch.from_numpy(S.synthetic_frames(B, cfg, 0)).cuda()
mask = torch.from_numpy(S.synthetic_masks(B, cfg, kv, 0, clump)).cuda()
nv = cfg.tokens_per_frame + kv
lib = _lib.get_dev_lib()
m.predict_video(x, mask, n_vis=nv)
m.set_lanes(lanes)
DEFAULTS = {"gemm_debug": 0, "gemm_tile": 0, "attn_ksplit": 1, "attn_tail": 1}
def run():
    for _ in range(5): m.predict_video(x, mask, n_vis=nv, check=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m.predict_video(x, mask, n_vis=nv, check=False)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 20
for rep in range(3):
    for st, label in zip(settings, sys.argv[2:]):
        for k in keys: m.set_option(k, DEFAULTS.get(k, 0))
        for k, v in st: m.set_option(k, int(v))
        dt = run()
        print("%-36s lanes %d: %.3f ms/step  %.0f frames/s" % (label, lanes, 1e3 * dt, B / dt), flush=True)
