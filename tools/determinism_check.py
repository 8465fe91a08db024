"""Run-to-run determinism of the forward (development tool, GPU box): N forwards of one batch, compared by the sum of their outputs.

    CFG=large_4x4patch_2frames_1tube BATCH=8 CALLS=300 python tools/determinism_check.py     # the workload that exposed the LDS-wait race of round 4

(MODE=fast, SIDE_STREAM=1: the call runs on a non-default torch stream.)"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch
from counterfactualworldmodels_amd import _lib, config as C, synthetic as S, vmae
from oracle import vmae_oracle as O
if os.environ.get("CFG") == "imu":
    import math
    from counterfactualworldmodels_amd import conjoined_vmae as CV
    cfgc = C.CONJ_CONFIGS["imu400_base_4x4patch_2frames_1tube"]
    mc_ = CV.ConjoinedPaddedVisionTransformer(cfgc, mode=os.environ.get("MODE", "parity"))
    mc_.load_state_dict({k: torch.from_numpy(S.synthetic_tensor(k, shp, 0)) for k, shp in C.conj_state_dict_schema(cfgc).items()})
    mc_ = mc_.cuda().eval()
    B = int(os.environ.get("BATCH", 16))
    xi = torch.from_numpy(S.synthetic_frames(B, cfgc.main, 0)).cuda().transpose(1, 2)
    mi = torch.from_numpy(S.synthetic_masks(B, cfgc.main, 4, 0)).cuda()
    imu = (torch.randn(B, 6, 400, generator=torch.Generator().manual_seed(0)) * 0.1).cuda()
    mctx = torch.zeros(B, 25, dtype=torch.bool, device="cuda")
    step = lambda: mc_(xi, mi, x_context=imu, mask_context=mctx, normalize=True, check=False)
    step()
    N = int(os.environ.get("CALLS", 60))
    sums = [float(step().double().sum()) for _ in range(N)]
    fin = [s_ for s_ in sums if math.isfinite(s_)]
    ref = max(set(fin), key=fin.count)
    print("%s: %d of %d calls differ (%d of them not finite)" % (os.environ.get("TAG", "imu"), sum(s_ != ref for s_ in sums), N, N - len(fin)))
    sys.exit(0)
cfg = C.CONFIGS[os.environ.get("CFG", "large_4x4patch_2frames_1tube")]
kv, clump = (8, 1) if "base" in cfg.name else (32, 2)
m = vmae.PretrainVisionTransformer(cfg, mode=os.environ.get("MODE", "parity"))
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
B = int(os.environ.get("BATCH", 8)); n_vis = cfg.tokens_per_frame + kv
x = O.preprocess(torch.from_numpy(S.synthetic_frames(B, cfg, 0))).cuda()
mask = torch.from_numpy(S.synthetic_masks(B, cfg, kv, 0, clump)).cuda()
m(x, mask, n_vis=n_vis)
N = int(os.environ.get("CALLS", 150))
import math
if os.environ.get("SIDE_STREAM"):
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        m(x, mask, n_vis=n_vis)
        sums = [float(m(x, mask, n_vis=n_vis).double().sum()) for _ in range(N)]
else:
    sums = [float(m(x, mask, n_vis=n_vis).double().sum()) for _ in range(N)]
fin = [s_ for s_ in sums if math.isfinite(s_)]
ref = max(set(fin), key=fin.count)
print("%s: %d of %d calls differ (%d of them not finite)" % (os.environ.get("TAG", ""), sum(s_ != ref for s_ in sums), N, N - len(fin)))
