"""Per-operator precision budget of the predictor forward pass, measured on the CPU oracle (no GPU).

    python tests/precision_budget.py [--cases base8,sharp,large4] [--out profiles/r3_precision_budget.txt]

For every class of matrix product on the path (q k^T, P v, qkv, proj, fc1, fc2) the operands of THAT class alone are rounded
the way one MFMA scheme would see them (oracle/vmae_oracle.py `PRECISION`), everything else stays fp32, and the max-abs error of
the model output against the reference's golden output is recorded.  The HIP library's `parity` mode is "bf16x3" on every class;
the table says which classes can drop to fewer MFMAs per product inside the 1e-3 budget.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from counterfactualworldmodels_amd import config as C, synthetic as S  # noqa: E402
from oracle import vmae_oracle as O  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = {
    "base8": ("base8_k8_b2.npz", "base_8x8patch_2frames_1tube", False),
    "sharp": ("base8_sharp_b1.npz", "base_8x8patch_2frames_1tube", True),
    "large4": ("large4_k32_b1.npz", "large_4x4patch_2frames_1tube", False),
}
CLASSES = ["qk", "pv", "qkv", "proj", "fc1", "fc2"]
SCHEMES = ["fp16", "fp16_a2", "fp16_b2", "bf16x3", "bf16"]


def run(case, table):
    name, cfg_name, sharp = CASES[case]
    g = np.load(os.path.join(GOLDEN, name))
    cfg = C.CONFIGS[cfg_name]
    seed, batch, k_vis, clump = int(g["seed"]), int(g["batch"]), int(g["k_vis"]), int(g["clump"])
    W = {k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, seed, sharp=sharp).items()}
    x = O.preprocess(torch.from_numpy(S.synthetic_frames(batch, cfg, seed)))
    mask = torch.from_numpy(S.synthetic_masks(batch, cfg, k_vis, seed, clump))
    O.PRECISION.clear()
    O.PRECISION.update(table)
    t0 = time.time()
    with torch.no_grad():
        y = O.vmae_forward(W, O.SPECS[cfg_name], x, mask).numpy()
    O.PRECISION.clear()
    d = np.abs(y - g["y_tokens"])
    return float(d.max()), float(d.mean()), time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="base8,sharp,large4")
    ap.add_argument("--out", default=None)
    ap.add_argument("--combos-only", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)
        if args.out:
            with open(args.out, "w") as f:
                f.write("\n".join(lines) + "\n")

    cases = args.cases.split(",")
    emit("# max-abs (mean-abs) error of the model output vs the reference golden; operands of ONE class rounded, all others fp32")
    emit("# %-28s " % "table" + " ".join("%-22s" % c for c in cases))
    tables = [("fp32 (oracle itself)", {})]
    if not args.combos_only:
        for cls in CLASSES:
            for sch in SCHEMES:
                tables.append((f"{cls}={sch}", {cls: sch}))
    par = {c: "bf16x3" for c in CLASSES}
    tables += [
        ("all=bf16x3 (parity)", dict(par)),
        ("parity, qk=pv=fp16", dict(par, qk="fp16", pv="fp16")),
        ("parity, pv=fp16", dict(par, pv="fp16")),
        ("parity, qk=fp16_a2 pv=fp16", dict(par, qk="fp16_a2", pv="fp16")),
        ("parity, qk=fp16x3 pv=fp16", dict(par, qk="fp16x3", pv="fp16")),
        ("parity, pv=bf16", dict(par, pv="bf16")),
        ("all=fp16", {c: "fp16" for c in CLASSES}),
        ("all=bf16 (fast)", {c: "bf16" for c in CLASSES}),
    ]
    for label, table in tables:
        cells = []
        for case in cases:
            mx, mean, dt = run(case, table)
            cells.append("%.2e (%.1e) %3.0fs" % (mx, mean, dt))
        emit("%-30s " % label + " ".join("%-22s" % c for c in cells))


if __name__ == "__main__":
    main()
