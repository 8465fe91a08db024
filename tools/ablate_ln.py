"""Upper bound of any LayerNorm fusion: the bench step with every LayerNorm launch skipped (results are wrong, timing only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import _lib, config as C, synthetic as S, vmae
cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
m = vmae.PretrainVisionTransformer(cfg, mode="parity")
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
x = torch.from_numpy(S.synthetic_frames(32, cfg, 0)).cuda()
mask = torch.from_numpy(S.synthetic_masks(32, cfg, 8, 0)).cuda()
lib = _lib.get_lib()
def run(tag):
    for _ in range(5): m.predict_video(x, mask, n_vis=792, check=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m.predict_video(x, mask, n_vis=792, check=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("%-28s %.2f ms/step  %.0f frames/s" % (tag, 1e3 * dt, 32 / dt), flush=True)
_lib.check(lib.cwm_debug_set(b"ln_fuse", 1))  # fold state is allocated at model creation, only while the switch is on
m.predict_video(x, mask, n_vis=792)
for lanes in (2, 1):
    m.set_lanes(lanes)
    for fuse, dbg, tag in ((1, 0, "LayerNorm folded into GEMMs"), (0, 0, "stand-alone LayerNorm"), (0, 8, "LayerNorm launches skipped"),
                           (1, 0, "LayerNorm folded into GEMMs"), (0, 0, "stand-alone LayerNorm")):
        _lib.check(lib.cwm_debug_set(b"ln_fuse", fuse))
        _lib.check(lib.cwm_debug_set(b"gemm_debug", dbg))
        run("lanes %d, %s" % (lanes, tag))
_lib.check(lib.cwm_debug_set(b"ln_fuse", 1))
_lib.check(lib.cwm_debug_set(b"gemm_debug", 0))
