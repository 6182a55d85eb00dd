"""Gaps in a rocprofv3 kernel trace: usage gap_report.py <kernel_trace.csv> [last_ms=130]
prints busy time, the idle gaps above 100 us and what ran before / after each."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 130.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]) for r in rows)
t_end = max(e[1] for e in ev)
ev = [e for e in ev if e[0] >= t_end - last_ms * 1e6]
busy = 0; cur_end = ev[0][0]; gaps = []
prev = None
for s, e, n in ev:
    if s > cur_end:
        gaps.append((s - cur_end, cur_end - ev[0][0], prev, n))
        busy += e - s; cur_end = e
    else:
        if e > cur_end: busy += e - cur_end; cur_end = e
    prev = n
span = cur_end - ev[0][0]
print("kernels %d, span %.1f ms, busy %.1f ms (%.0f%%), sum of durations %.1f ms" % (len(ev), span / 1e6, busy / 1e6, 100.0 * busy / span, sum(e - s for s, e, _ in ev) / 1e6))
print("gaps > 100 us: %d, total %.1f ms; all gaps total %.1f ms" % (sum(1 for g in gaps if g[0] > 1e5), sum(g[0] for g in gaps if g[0] > 1e5) / 1e6, sum(g[0] for g in gaps) / 1e6))
for g in gaps:
    if g[0] > 1e5: print("  at %7.2f ms  idle %7.1f us   after %-50s before %s" % (g[1] / 1e6, g[0] / 1e3, g[2], g[3]))
