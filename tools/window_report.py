"""The kernels of the LAST `last_ms` milliseconds of a rocprofv3 kernel trace, by name: usage window_report.py <kernel_trace.csv> [last_ms=75]
-> busy time (union of intervals), sum of durations, and per kernel: calls, summed duration, the time it was the ONLY kernel on the device"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 75.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zk::", "")[:64]) for r in rows)
t_end = max(e[1] for e in ev)
ev = [e for e in ev if e[0] >= t_end - last_ms * 1e6]
# sweep: at every instant, which kernels run
pts = []
for i, (s, e, n) in enumerate(ev):
    pts.append((s, 1, i)); pts.append((e, 0, i))
pts.sort()
active, last_t = set(), pts[0][0]
alone = collections.Counter(); busy = 0; conc_time = collections.Counter()
for t, kind, i in pts:
    if active:
        busy += t - last_t
        conc_time[min(len(active), 8)] += t - last_t
        if len(active) == 1: alone[ev[next(iter(active))][2]] += t - last_t
    last_t = t
    if kind: active.add(i)
    else: active.discard(i)
span = ev[-1][1] - ev[0][0] if ev else 0
tot = collections.Counter(); calls = collections.Counter()
for s, e, n in ev: tot[n] += e - s; calls[n] += 1
print("window %.1f ms: %d kernels, busy %.1f ms (%.0f%%), sum of durations %.1f ms" % (span / 1e6, len(ev), busy / 1e6, 100.0 * busy / max(span, 1), sum(tot.values()) / 1e6))
print("time with k kernels on the device: " + ", ".join("%d%s: %.1f ms" % (k, "+" if k == 8 else "", v / 1e6) for k, v in sorted(conc_time.items())))
print("%-66s %6s %10s %10s" % ("kernel", "calls", "sum ms", "alone ms"))
for n, v in tot.most_common(24): print("%-66s %6d %10.2f %10.2f" % (n, calls[n], v / 1e6, alone[n] / 1e6))
