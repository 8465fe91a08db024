"""Host enqueue time of one IMU-conditioned forward against its GPU time (batch 16, parity):  python tools/conj_host_issue.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counterfactualworldmodels_amd import _lib, config as C, synthetic as S, conjoined_vmae as CV
cfg = C.CONJ_CONFIGS["imu400_base_4x4patch_2frames_1tube"]
B = int(os.environ.get("BATCH", 16))
m = CV.ConjoinedPaddedVisionTransformer(cfg, mode="parity")
m.load_state_dict({k: torch.from_numpy(S.synthetic_tensor(k, shp, 0)) for k, shp in C.conj_state_dict_schema(cfg).items()})
m = m.cuda().eval()
x = torch.from_numpy(S.synthetic_frames(B, cfg.main, 0)).cuda().transpose(1, 2)
mask = torch.from_numpy(S.synthetic_masks(B, cfg.main, 4, 0)).cuda()
imu = (torch.randn(B, 6, 400, generator=torch.Generator().manual_seed(0)) * 0.1).cuda()
mc = torch.zeros(B, 25, dtype=torch.bool, device="cuda")
step = lambda: m(x, mask, x_context=imu, mask_context=mc, normalize=True, check=False)
step(); step()
lib = _lib.get_dev_lib()
for lanes in (2, 1):
    m.set_lanes(lanes)
    for ctx in (1, 0):
        m.set_option("conj_ctx_stream", ctx)
        step(); torch.cuda.synchronize()
        n = 6
        t0 = time.perf_counter()
        for _ in range(n): step()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        # a single call with an empty queue: host time of the call itself
        torch.cuda.synchronize(); t3 = time.perf_counter(); step(); t4 = time.perf_counter(); torch.cuda.synchronize(); t5 = time.perf_counter()
        print("lanes %d ctx_stream %d: %.2f ms/step; host issue of %d queued calls %.2f ms each; one call on an empty queue: host %.2f ms, GPU done after %.2f ms" % (
            lanes, ctx, 1e3 * (t2 - t0) / n, n, 1e3 * (t1 - t0) / n, 1e3 * (t4 - t3), 1e3 * (t5 - t3)), flush=True)
m.set_option("conj_ctx_stream", 1)
