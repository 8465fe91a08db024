"""Kernel trace of a few wrapper steps (run under rocprofv3 --kernel-trace): the idle time between the last kernel of a step and the first of the next
(the wrapper's one host read-back per call, RectangularizeMasks).   rocprofv3 --kernel-trace -d OUT -- python3 tools/wrap_gap.py ; then
python3 tools/wrap_gap.py OUT  prints the gaps."""
import glob, os, sys
if len(sys.argv) > 1:
    import csv
    f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = 0
    for i, r in enumerate(rows):
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if "unembed" in r["Kernel_Name"] and i + 1 < len(rows):
            # last un-embed of the step = the one followed by a non-lane kernel
            nxt = rows[i + 1]
            gap = (int(nxt["Start_Timestamp"]) - max(ends, en)) / 1e3
            if gap > 30:
                print("step end -> next kernel %-40s gap %.1f us" % (nxt["Kernel_Name"][:40], gap))
                t0 = max(ends, en)
                for r2 in rows[i + 1 : i + 9]:
                    print("      +%7.1f us .. +%7.1f us  %s" % ((int(r2["Start_Timestamp"]) - t0) / 1e3, (int(r2["End_Timestamp"]) - t0) / 1e3, r2["Kernel_Name"][:60]))
        ends = max(ends, en)
    sys.exit(0)
sys.path.insert(0, os.getcwd())
import torch
from counterfactualworldmodels_amd import config as C, synthetic as S, vmae, segmentation
cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
m = vmae.PretrainVisionTransformer(cfg, mode="parity")
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
x = torch.from_numpy(S.synthetic_frames(32, cfg, 0)).cuda()
mask = torch.from_numpy(S.synthetic_masks(32, cfg, 8, 0, 1)).cuda()
G = segmentation.FlowGenerator(predictor=m, imagenet_normalize_inputs=True, temporal_dim=2)
for _ in range(8):
    G.predict(x, mask, frame=None)
torch.cuda.synchronize()
