"""Same-box A/B of two builds of the library (same C ABI; side builds: CWM_HIP_LIB_OUT=... python -m counterfactualworldmodels_amd.build, which also writes
<name>_dev.so): python tools/ab_lib.py LIB_A LIB_B [gemm|attn|step ...]  (each measured in its own child process,
alternating A B A B).  gemm: the ViT-B/8 GEMM shapes, parity mode, default tile rule; attn: the attention shapes, both modes; step: the
ViT-B/8 batch-32 step on two lanes and on one."""
import os
import subprocess
import sys

CHILD = r'''
import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from counterfactualworldmodels_amd import _lib, config as CF, synthetic as S, vmae
from tools.microbench import ATTN_SHAPES, GEMM_SHAPES
what = sys.argv[1].split(",")
torch.cuda.init()
lib = _lib.get_dev_lib(); us = C.c_double()
if "gemm" in what:
    for name, M, N, K, epi in GEMM_SHAPES:
        best = 1e30
        for _ in range(3):
            _lib.check(lib.cwm_bench_gemm(M, N, K, _lib.mode_id("parity"), epi, 20, C.byref(us)))
            best = min(best, us.value)
        print("gemm %-12s parity %8.1f us %6.1f TF" % (name, best, 2.0 * M * N * K / best / 1e6), flush=True)
if "attn" in what:
    for name, B, H, N in ATTN_SHAPES + [("b8.enc.half", 16, 12, 792), ("b8.dec.half", 16, 6, 1568)]:
        for mode in ("parity", "fast"):
            best = 1e30
            for _ in range(3):
                _lib.check(lib.cwm_bench_attention(B, H, N, _lib.mode_id(mode), 30, C.byref(us)))
                best = min(best, us.value)
            print("attn %-12s %-6s %8.1f us %6.1f TF" % (name, mode, best, 4.0 * N * N * 64 * B * H / best / 1e6), flush=True)
if "step" in what:
    cfg = CF.CONFIGS["base_8x8patch_2frames_1tube"]
    m = vmae.PretrainVisionTransformer(cfg, mode="parity")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
    m = m.cuda().eval()
    x = torch.from_numpy(S.synthetic_frames(32, cfg, 0)).cuda()
    mask = torch.from_numpy(S.synthetic_masks(32, cfg, 8, 0, 1)).cuda()
    m.predict_video(x, mask, n_vis=792)
    for lanes in (2, 1):
        m.set_lanes(lanes)
        best = 1e30
        for _ in range(3):
            for _ in range(5): m.predict_video(x, mask, n_vis=792, check=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): m.predict_video(x, mask, n_vis=792, check=False)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20)
        print("step lanes %d %8.3f ms %7.1f frames/s" % (lanes, 1e3 * best, 32 / best), flush=True)
'''
what = ",".join(sys.argv[3:]) or "gemm,step"
for rep in range(2):
    for tag, lib in (("A", sys.argv[1]), ("B", sys.argv[2])):
        env = dict(os.environ, CWM_HIP_LIB=os.path.abspath(lib))  # (its development twin <name>_dev.so is found beside it: _lib.dev_library_path)
        out = subprocess.run([sys.executable, "-c", CHILD, what], env=env, capture_output=True, text=True)
        for line in out.stdout.strip().split("\n"):
            print("%s%d %-22s %s" % (tag, rep, os.path.basename(lib), line), flush=True)
        if out.returncode:
            print(out.stderr[-2000:])
