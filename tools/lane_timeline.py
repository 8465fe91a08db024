"""Two-lane timeline from a rocprofv3 kernel trace of the default bench command (tools/collect_profiles.sh: stats_base8_parity_default):
for the timed two-lane region -- per kernel class: launches, mean duration against the one-lane region's, and how much of the time the chip ran
kernels of 0 / 1 / 2 queues at once.   python tools/lane_timeline.py TRACE.csv"""
import csv, re, sys
from collections import defaultdict
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
for r in rows:
    r["s"], r["e"], r["q"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    n = r["Kernel_Name"]
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n); r["k"] = n.replace("cwm::", "")
rows.sort(key=lambda r: r["s"])
# steps = un-embed launches; a two-lane step has its kernels on two queues
steps, cur = [], []
for r in rows:
    cur.append(r)
    if r["k"].startswith("unembed3"):
        pass
# split into steps at mask_to_perm of the first lane: a step starts at the first mask_to_perm after an unembed
bounds = [i for i, r in enumerate(rows) if r["k"].startswith("mask_to_perm")]
starts = []
last_unembed = -1
for i, r in enumerate(rows):
    if r["k"].startswith("unembed3"): last_unembed = i
    if r["k"].startswith("mask_to_perm") and (not starts or last_unembed > starts[-1]): starts.append(i)
starts.append(len(rows))
two, one = [], []
for a, b in zip(starts[:-1], starts[1:]):
    seg = [r for r in rows[a:b] if not r["k"].startswith("__amd") and "at::" not in r["k"]]
    if not seg: continue
    qs = {r["q"] for r in seg}
    (two if len(qs) >= 2 else one).append(seg)
print("steps: %d on two queues, %d on one" % (len(two), len(one)))
def stats(segs):
    d = defaultdict(list)
    for seg in segs:
        for r in seg: d[r["k"]].append((r["e"] - r["s"]) / 1e3)
    return d
d2, d1 = stats(two[1:]), stats(one[1:])
print("%-46s %6s %10s %10s %7s" % ("kernel", "n/step", "us 2 lanes", "us 1 lane", "ratio"))
tot2 = tot1 = 0.0
for k in sorted(d2, key=lambda k: -sum(d2[k])):
    n2 = len(d2[k]) / max(len(two) - 1, 1); m2 = sum(d2[k]) / len(d2[k]); m1 = sum(d1[k]) / len(d1[k]) if d1.get(k) else float("nan")
    tot2 += sum(d2[k]) / max(len(two) - 1, 1); tot1 += (sum(d1[k]) / max(len(one) - 1, 1)) if d1.get(k) else 0
    if sum(d2[k]) / max(len(two) - 1, 1) > 20: print("%-46s %6.1f %10.1f %10.1f %7.2f" % (k[:46], n2, m2, m1, m2 / m1 if m1 == m1 else float("nan")))
print("sum of kernel durations per step: two lanes %.2f ms, one lane %.2f ms" % (tot2 / 1e3, tot1 / 1e3))
for name, segs in (("two lanes", two[1:]), ("one lane", one[1:])):
    occ = defaultdict(float); span = 0.0
    for seg in segs:
        ev = []
        for r in seg: ev += [(r["s"], 1), (r["e"], -1)]
        ev.sort()
        lvl, t0 = 0, ev[0][0]
        for t, dl in ev:
            occ[min(lvl, 3)] += (t - t0) / 1e3; t0 = t; lvl += dl
        span += (ev[-1][0] - ev[0][0]) / 1e3
    print("%s: step span %.2f ms; time with 0 / 1 / 2 / 3+ kernels in flight: %s" % (name, span / len(segs) / 1e3, " / ".join("%.2f" % (occ[i] / len(segs) / 1e3) for i in range(4))))
