"""Summarise a tools/collect_profiles.sh output directory (gpurun_out/prof_<tag>) into profiles/ (run in the build container):
    python tools/summarize_profiles.py r4
  profiles/<tag>_kernel_stats_bench_<workload>_<mode>.csv      rocprofv3 --stats of bench.py --lanes 1 (and *_parity_default: two lanes)
  profiles/<tag>_bench_under_rocprof_<workload>_parity.json    the bench line of that very run (its HIP-event averages must agree)
  profiles/<tag>_pmc_summary.json                              per workload and kernel: HBM bytes per launch, MFMA busy, clock
  profiles/pmc_summary_latest.json                             what bench.py reads for roofline.traffic / mfma_busy
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced stream (MI355X_MICROARCH.md, HBM
section) and is doubled here; WRITE_SIZE is exact for 16-byte-per-lane stores.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE /
8 XCDs x 1024 SIMDs)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
WORKLOADS = ("base8", "large4", "imu4")


def one(pattern):
    """The NEWEST match: a second collection into the same tag leaves the first one's files (other process ids) beside its own."""
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(f, key=os.path.getmtime) if f else None


def kname(raw):
    return raw.split("(")[0].replace("void ", "").strip()


def counters(d):
    """{(kernel, counter): [per-dispatch values]}, {kernel: {dispatch: ns}} of one PMC pass."""
    f = one(d + "/**/*counter_collection.csv")
    agg, dur = collections.defaultdict(list), collections.defaultdict(dict)
    if not f:
        return agg, dur
    for r in csv.DictReader(open(f)):
        k = kname(r["Kernel_Name"])
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg, dur


for wl in WORKLOADS:
    for mode in ("parity", "fast", "parity_default"):
        f = one("stats_%s_%s/**/*kernel_stats.csv" % (wl, mode))
        if f:
            shutil.copy(f, os.path.join(dst, "%s_kernel_stats_bench_%s_%s.csv" % (tag, wl, mode)))
    f = os.path.join(src, "bench_under_rocprof_%s_parity.json" % wl)
    if os.path.exists(f) and os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(dst, "%s_bench_under_rocprof_%s_parity.json" % (tag, wl)))
    else:
        print("WARNING: no bench line captured under rocprofv3 for", wl)
f = one("stats_b1/**/*kernel_stats.csv")
if f:
    shutil.copy(f, os.path.join(dst, "%s_kernel_stats_latency_b1_parity.csv" % tag))

out = {"tag": tag, "command": "python3 bench.py --workload <w> --steps 5|3 --warmup 1 --no-cpu-baseline --no-secondary --no-prompts --lanes 1 (parity mode; "
                              "imu4 with CWM_CONJ_CTX_STREAM=0: no two kernels overlap)", "workloads": {}}
latest = {"entries": {}}
for wl in WORKLOADS:
    kernels = {}
    for d, name in (("pmc_fetch_" + wl, "FETCH_SIZE"), ("pmc_write_" + wl, "WRITE_SIZE")):
        agg, _ = counters(d)
        for (k, c), v in agg.items():
            if c == name and "cwm::" in k:
                t = kernels.setdefault(k, {})
                t["launches"] = len(v)
                t[name + "_KiB_mean_raw"] = sum(v) / len(v)
    for k, t in kernels.items():
        t["hbm_bytes_per_launch_corrected"] = (2.0 * t.get("FETCH_SIZE_KiB_mean_raw", 0.0) + t.get("WRITE_SIZE_KiB_mean_raw", 0.0)) * 1024.0
    agg, dur = counters("pmc_mfma_" + wl)
    tot = collections.defaultdict(dict)
    for (k, c), v in agg.items():
        if "cwm::" in k:
            tot[k][c] = sum(v)
    for k, d in tot.items():
        t = kernels.setdefault(k, {})
        if d.get("GRBM_GUI_ACTIVE"):
            cycles = d["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
            ns = sum(dur[k].values())
            t.update(mfma_busy=d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cycles * 1024.0), clock_GHz=cycles / ns if ns else None,
                     total_us_in_pmc_pass=ns / 1e3, SQ_WAIT_ANY_share=(d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"]) if d.get("SQ_WAVE_CYCLES") else None)
    if not kernels:
        continue
    out["workloads"][wl] = dict(sorted(kernels.items(), key=lambda kv: -kv[1].get("total_us_in_pmc_pass", 0.0)))
    latest["entries"]["%s/parity" % wl] = {
        "tag": tag, "source": "profiles/%s_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES, separate passes)" % tag,
        "hbm_bytes_per_launch": {k: t["hbm_bytes_per_launch_corrected"] for k, t in kernels.items() if t.get("hbm_bytes_per_launch_corrected")},
        "mfma_busy": {k: t["mfma_busy"] for k, t in kernels.items() if t.get("mfma_busy")}}

# the edge workloads (bench.py --workload flowstats / prompt_build): kernel stats, the bench line of that run, HBM bytes per launch of every kernel
for wl in ("flowstats", "prompt_build"):
    f = one("stats_%s_f32/**/*kernel_stats.csv" % wl)
    if f:
        shutil.copy(f, os.path.join(dst, "%s_kernel_stats_bench_%s.csv" % (tag, wl)))
    f = os.path.join(src, "bench_under_rocprof_%s_f32.json" % wl)
    if os.path.exists(f) and os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(dst, "%s_bench_under_rocprof_%s.json" % (tag, wl)))
    kernels = {}
    for d, name in (("pmc_fetch_" + wl, "FETCH_SIZE"), ("pmc_write_" + wl, "WRITE_SIZE")):
        agg, dur = counters(d)
        for (k, c), v in agg.items():
            if c == name and "cwm::" in k:
                t = kernels.setdefault(k, {})
                t["launches"] = len(v)
                t[name + "_KiB_mean_raw"] = sum(v) / len(v)
                # (these workloads launch one kernel at several sizes -- S = 256 and S = 24 -- so the LARGEST dispatches are reported beside the mean)
                big = sorted(v)[-max(1, len(v) // 4):]
                t[name + "_KiB_top_quartile_raw"] = sum(big) / len(big)
                t["avg_us_in_pmc_pass"] = sum(dur[k].values()) / max(len(dur[k]), 1) / 1e3
    for k, t in kernels.items():
        t["hbm_bytes_per_launch_corrected"] = (2.0 * t.get("FETCH_SIZE_KiB_mean_raw", 0.0) + t.get("WRITE_SIZE_KiB_mean_raw", 0.0)) * 1024.0
        t["hbm_bytes_per_launch_corrected_top_quartile"] = (2.0 * t.get("FETCH_SIZE_KiB_top_quartile_raw", 0.0) + t.get("WRITE_SIZE_KiB_top_quartile_raw", 0.0)) * 1024.0
    if wl == "flowstats":  # the covariance kernel on the fp32 matrix pipe: busy share and clock per dispatch, split by duration (S = 256 launches vs S = 24 launches)
        agg, dur = counters("pmc_mfma_flowstats")
        f = one("pmc_mfma_flowstats/**/*counter_collection.csv")
        if f:
            per = collections.defaultdict(dict)
            for r in csv.DictReader(open(f)):
                if "flow_cov_kernel" in r["Kernel_Name"]:
                    per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
                    per[r["Dispatch_Id"]]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                    per[r["Dispatch_Id"]]["kernel"] = kname(r["Kernel_Name"])
            groups = collections.defaultdict(list)
            for d in per.values():
                if d.get("GRBM_GUI_ACTIVE"):
                    cyc = d["GRBM_GUI_ACTIVE"] / 8.0
                    groups[d["kernel"]].append((d["ns"] / 1e3, cyc / d["ns"], d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0), d.get("SQ_WAIT_ANY", 0.0) / max(d.get("SQ_WAVE_CYCLES", 1.0), 1.0)))
            for k, v in groups.items():
                t = kernels.setdefault(k, {})
                n = float(len(v))
                t.update(dispatches_in_mfma_pass=len(v), avg_us_in_mfma_pass=sum(x[0] for x in v) / n, clock_GHz=sum(x[1] for x in v) / n, mfma_busy=sum(x[2] for x in v) / n,
                         SQ_WAIT_ANY_share=sum(x[3] for x in v) / n)
    if kernels:
        out["workloads"][wl] = kernels
        latest["entries"]["%s/f32" % wl] = {"tag": tag, "source": "profiles/%s_pmc_summary.json" % tag,
                                            "hbm_bytes_per_launch": {k: t["hbm_bytes_per_launch_corrected_top_quartile"] for k, t in kernels.items() if "hbm_bytes_per_launch_corrected_top_quartile" in t},
                                            "mfma_busy": {k: t["mfma_busy"] for k, t in kernels.items() if "mfma_busy" in t}}

for mode in ("fast", "parity"):
    agg, dur = counters("pmc_attn_l4dec_" + mode)
    # (the key-split tail round adds attention_combine_kernel dispatches: the measured kernel is the main one)
    d = {c: v[-1] for (k, c), v in agg.items() if "attention" in k and "combine" not in k}
    durs = [v for k, v in dur.items() if "attention" in k and "combine" not in k]
    if d and durs:
        ns = list(durs[0].values())[-1]
        cycles = d["GRBM_GUI_ACTIVE"] / 8.0
        d.update(duration_us=ns / 1e3, clock_GHz=cycles / ns, mfma_busy=d["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0),
                 algorithmic_TFLOPs=4.0 * 8 * 8 * 6272 * 6272 * 64 / ns / 1e3)
        out["attention_vit_l4_decoder_B8_H8_N6272_" + mode] = d
json.dump(out, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1)
if latest["entries"]:
    json.dump(latest, open(os.path.join(dst, "pmc_summary_latest.json"), "w"), indent=1)
for wl, ks in out["workloads"].items():
    for k, t in ks.items():
        print("%-7s %-60s launches %4s  HBM %7.1f MB/launch  mfma busy %s  clock %s" % (
            wl, k[:60], t.get("launches"), t.get("hbm_bytes_per_launch_corrected", 0) / 1e6,
            "%.3f" % t["mfma_busy"] if t.get("mfma_busy") is not None else "-", "%.2f" % t["clock_GHz"] if t.get("clock_GHz") else "-"))
for mode in ("fast", "parity"):
    d = out.get("attention_vit_l4_decoder_B8_H8_N6272_" + mode)
    if d:
        print("L/4 decoder attention %-6s %.1f us  %.1f TFLOP/s  mfma busy %.3f" % (mode, d["duration_us"], d["algorithmic_TFLOPs"], d["mfma_busy"]))
