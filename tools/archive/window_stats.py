"""Per-kernel totals of the LAST `ms` milliseconds of a rocprofv3 kernel trace: usage window_stats.py <kernel_trace.csv> [ms=80]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 80.0
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zk::", "")[:70]) for r in rows]
t_end = max(e[1] for e in ev)
ev = sorted(e for e in ev if e[0] >= t_end - ms * 1e6)
tot = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev:
    tot[n][0] += 1; tot[n][1] += e - s
pts = sorted([(s, 1) for s, e, n in ev] + [(e, -1) for s, e, n in ev])
cur = 0; last = None; union = 0
for t, d in pts:
    if last is not None and cur > 0: union += t - last
    cur += d; last = t
span = ev[-1][1] - ev[0][0]
print("window %.1f ms: %d kernels, sum of durations %.1f ms, GPU busy (union) %.1f ms of %.1f ms span" % (ms, len(ev), sum(v[1] for v in tot.values()) / 1e6, union / 1e6, span / 1e6))
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%6d calls %9.2f ms total %9.1f us avg  %s" % (c, t / 1e6, t / 1e3 / c, n))
