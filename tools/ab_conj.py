"""Same-box A/B of the IMU-conditioned model's batch-16 step for one development switch:  python tools/ab_conj.py KEY VALUE_A VALUE_B [lanes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import _lib, config as C, synthetic as S, conjoined_vmae as CV
key, va, vb = sys.argv[1].encode(), int(sys.argv[2]), int(sys.argv[3])
lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 2
cfg = C.CONJ_CONFIGS["imu400_base_4x4patch_2frames_1tube"]
B = 16
m = CV.ConjoinedPaddedVisionTransformer(cfg, mode=os.environ.get("MODE", "parity"))
m.load_state_dict({k: torch.from_numpy(S.synthetic_tensor(k, shp, 0)) for k, shp in C.conj_state_dict_schema(cfg).items()})
m = m.cuda().eval()
x = torch.from_numpy(S.synthetic_frames(B, cfg.main, 0)).cuda().transpose(1, 2)
mask = torch.from_numpy(S.synthetic_masks(B, cfg.main, 4, 0)).cuda()
imu = (torch.randn(B, 6, 400, generator=torch.Generator().manual_seed(0)) * 0.1).cuda()
mc = torch.zeros(B, 25, dtype=torch.bool, device="cuda")
lib = _lib.get_dev_lib()
step = lambda: m(x, mask, x_context=imu, mask_context=mc, normalize=True, check=False)
step()
m.set_lanes(lanes)
def run():
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 6
ys = {}
for rep in range(3):
    for v in (va, vb):
        m.set_option(key.decode() if isinstance(key, bytes) else key, v)
        dt = run()
        ys[v] = step()
        print("%s=%d lanes %d: %.3f ms/step  %.1f frames/s" % (key.decode(), v, lanes, 1e3 * dt, B / dt), flush=True)
print("max |y(A) - y(B)| = %.3e" % (ys[va] - ys[vb]).abs().max().item())
