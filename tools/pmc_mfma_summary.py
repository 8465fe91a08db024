"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE):
util = busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); prints JSON {kernel: {launches, total_us, mfma_util, clock_GHz}}."""
import collections
import csv
import glob
import json
import os
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
agg, dur = collections.defaultdict(float), collections.defaultdict(dict)
for r in csv.DictReader(open(f[0])) if f else ():
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
    dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
out = {}
for k, d in dur.items():
    ns = sum(d.values())
    cyc = agg.get((k, "GRBM_GUI_ACTIVE"), 0.0) / 8.0
    busy = agg.get((k, "SQ_VALU_MFMA_BUSY_CYCLES"), 0.0)
    out[k] = {"launches": len(d), "total_us": ns / 1e3, "mfma_util": busy / (cyc * 1024.0) if cyc else None, "clock_GHz": cyc / ns if ns else None}
print(json.dumps(dict(sorted(out.items(), key=lambda kv: -kv[1]["total_us"])), indent=1))
