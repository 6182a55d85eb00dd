#!/usr/bin/env python3
"""every 2^21 x 256 leaf-hash dispatch of a run, one row each -- duration from `rocprofv3 --kernel-trace`, and from a
`--pmc GRBM_GUI_ACTIVE` pass of the same command its cycles and the clock they imply (GRBM_GUI_ACTIVE / 8 XCDs / duration of that pass's
dispatch; counter passes serialise the dispatches, so the two passes are different operating points and are shown side by side, not joined).
usage: hash_dispatch_report.py <trace dir> <pmc dir> [label]"""
import csv
import glob
import os
import sys

KEY = "hash_rows_vec_kernel"
LEAVES = 1 << 21


def rows(d, suffix):
    out = []
    for f in glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def stats(v):
    v = sorted(v)
    n = len(v)
    return "n %d, min %.0f, p10 %.0f, median %.0f, p90 %.0f, max %.0f, mean %.1f" % (n, v[0], v[n // 10], v[n // 2], v[(9 * n) // 10], v[-1], sum(v) / n) if n else "none"


trace_dir, pmc_dir = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else ""
print("# leaf-hash dispatches (`zk::hash_rows_vec_kernel`, 2^21 leaves), %s\n" % label)
tr = rows(trace_dir, "kernel_trace.csv")
allk = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)) for r in tr), key=lambda x: x[0])
big = [k for k in allk if KEY in k[2] and "batch" not in k[2] and k[3] == LEAVES and k[1] - k[0] >= 2000000]     # (>= 2 ms: the 256-column trace leaves; the 8-column quotient leaves of the same grid take 0.25 - 0.4 ms)
print("## kernel trace (dispatches overlap freely: four shards in flight)\n")
print("%d dispatches of the 256-column trace leaves; duration us: %s\n" % (len(big), stats([(e - s) / 1e3 for s, e, _, _ in big])))
solo = [(e - s) / 1e3 for i, (s, e, _, _) in enumerate(big) if not any(j != i and s2 < e and e2 > s for j, (s2, e2, _, _) in enumerate(big))]
print("of these, %d share no time with another leaf hash of that size (other shards' memory-bound and small kernels do run beside them); duration us: %s\n" % (len(solo), stats(solo)))
# how much of each hash dispatch's span other hash dispatches of that size share, and how many other kernels start inside it
print("| # | start ms | duration us | other big hashes overlapping (us) | other kernels starting inside | gap to previous big hash end (us) |")
print("|---|---|---|---|---|---|")
t0 = big[0][0] if big else 0
import bisect
starts = [k[0] for k in allk]
prev_end = None
for i, (s, e, _, _) in enumerate(big):
    ov = sum(max(0, min(e, e2) - max(s, s2)) for j, (s2, e2, _, _) in enumerate(big) if j != i and s2 < e and e2 > s) / 1e3
    inside = bisect.bisect_left(starts, e) - bisect.bisect_right(starts, s)
    gap = (s - prev_end) / 1e3 if prev_end is not None else float("nan")
    prev_end = e if prev_end is None else max(prev_end, e)
    print("| %d | %.2f | %.0f | %.0f | %d | %.0f |" % (i, (s - t0) / 1e6, (e - s) / 1e3, ov, inside, gap))
print()
pm = rows(pmc_dir, "counter_collection.csv")
by = {}
for r in pm:
    if KEY not in r["Kernel_Name"] or "batch" in r["Kernel_Name"] or int(r["Grid_Size"]) != LEAVES:
        continue
    d = by.setdefault(int(r["Dispatch_Id"]), {"v": 0.0, "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])})
    d["v"] += float(r["Counter_Value"])
by = {k: d for k, d in by.items() if d["e"] - d["s"] >= 2000000}
print("## `--pmc GRBM_GUI_ACTIVE` pass of the same command (dispatches serialised by the profiler)\n")
print("| # | duration us | GRBM_GUI_ACTIVE (sum of 8 XCDs) | clock GHz = cycles / 8 / duration |")
print("|---|---|---|---|")
clk = []
for i, (k, d) in enumerate(sorted(by.items())):
    us = (d["e"] - d["s"]) / 1e3
    ghz = d["v"] / 8 / (us * 1e3) if us else 0
    clk.append(ghz * 1e3)
    print("| %d | %.0f | %.4g | %.3f |" % (i, us, d["v"], ghz))
print("\nclock MHz over those dispatches: %s" % stats(clk))
