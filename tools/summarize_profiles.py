"""Summarise a tools/collect_profiles.sh output directory into profiles/<tag>_*.{csv,json} (run in the build container)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1d"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    f = glob.glob(os.path.join(src, pattern))
    return f[0] if f else None


for mode in ("parity", "fast", "parity_default"):
    f = one("stats_%s/runc/*kernel_stats.csv" % mode)
    if f:
        shutil.copy(f, os.path.join(dst, "%s_kernel_stats_bench_base8_%s.csv" % (tag, mode)))
f = one("stats_b1/runc/*kernel_stats.csv")
if f:
    shutil.copy(f, os.path.join(dst, "%s_kernel_stats_latency_b1_parity.csv" % tag))


def counters(d, kfilter):
    f = one(d + "/runc/*counter_collection.csv")
    agg, dur = collections.defaultdict(list), collections.defaultdict(dict)
    if not f:
        return agg, dur
    for r in csv.DictReader(open(f)):
        if kfilter in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg, dur


out = {"tag": tag, "command": "python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --no-prompts --lanes 1 (parity mode, B/8 batch 32; *_parity_default.csv: the same without --lanes 1)"}
# HBM traffic of the GEMM kernels: FETCH_SIZE / WRITE_SIZE are in KiB-units of 1024 B; on gfx950 FETCH_SIZE reports
# half of the bytes of a wide coalesced stream (MI355X_MICROARCH.md HBM section) -> doubled here.
traffic = {}
for d, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    agg, _ = counters(d, "gemm")
    for (k, c), v in agg.items():
        t = traffic.setdefault(k, {"launches": len(v)})
        t[name + "_KiB_mean_raw"] = sum(v) / len(v)
for k, t in traffic.items():
    t["hbm_bytes_per_launch_corrected"] = (2.0 * t.get("FETCH_SIZE_KiB_mean_raw", 0.0) + t.get("WRITE_SIZE_KiB_mean_raw", 0.0)) * 1024.0
out["gemm_hbm_traffic"] = traffic
tot_l = sum(t["launches"] for t in traffic.values()) or 1
out["gemm_hbm_bytes_per_launch_all"] = sum(t["hbm_bytes_per_launch_corrected"] * t["launches"] for t in traffic.values()) / tot_l

agg, dur = counters("pmc_mfma", "")
mf = {}
for (k, c), v in agg.items():
    mf.setdefault(k, {})[c] = sum(v)
for k, d in mf.items():
    if "GRBM_GUI_ACTIVE" in d and d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        cycles = d["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        d["mfma_util"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0)  # 256 CUs x 4 SIMDs
        ns = sum(dur[k].values())
        d["clock_GHz"] = cycles / ns if ns else None
out["bench_kernels_mfma"] = {k: v for k, v in mf.items() if "mfma_util" in v}

for mode in ("fast", "parity"):
    agg, dur = counters("pmc_attn_l4dec_" + mode, "attention")
    d = {}
    for (k, c), v in agg.items():
        d[c] = v[-1]
    if d:
        ns = list(list(dur.values())[0].values())[-1]
        cycles = d["GRBM_GUI_ACTIVE"] / 8.0
        d.update(duration_us=ns / 1e3, clock_GHz=cycles / ns, mfma_util=d["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0),
                 algorithmic_TFLOPs=4.0 * 8 * 8 * 6272 * 6272 * 64 / ns / 1e3)
        out["attention_vit_l4_decoder_B8_H8_N6272_" + mode] = d
json.dump(out, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1)
# what bench.py reads for roofline.traffic (same command: base8 workload, parity mode)
latest = {"tag": tag, "workload": "base8", "mode": "parity", "source": "profiles/%s_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % tag,
          "hbm_bytes_per_launch": {k: t["hbm_bytes_per_launch_corrected"] for k, t in traffic.items()}}
json.dump(latest, open(os.path.join(dst, "pmc_summary_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
