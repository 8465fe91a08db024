"""Timeline statistics of a rocprofv3 --kernel-trace run of bench.py: per kernel, time alone on the chip vs overlapped with the other
lane, and the idle time inside the steady-state window (the last third of the trace)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]) for r in csv.DictReader(open(f))]
rows.sort()
t_end = rows[-1][1]
span = t_end - rows[0][0]
lo = t_end - span // 3
win = [r for r in rows if r[0] >= lo]
events = []
for s, e, k in win:
    events.append((s, 1)); events.append((e, -1))
events.sort()
busy = over = 0; depth = 0; last = events[0][0]
for t, d in events:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d; last = t
total = win[-1][1] - win[0][0]
print("window %.2f ms: busy %.1f %%, two or more kernels in flight %.1f %%, idle %.1f %%" % (total / 1e6, 100 * busy / total, 100 * over / total, 100 * (total - busy) / total))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in win:
    agg[k][0] += 1; agg[k][1] += e - s
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    print("  %-50s %5d launches  mean %8.1f us  sum %7.2f ms (%.1f %% of window)" % (k, n, t / n / 1e3, t / 1e6, 100 * t / total))
