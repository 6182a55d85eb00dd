#!/usr/bin/env python3
"""Summarise the runs of tools/hash_insitu.py: gpurun_out/hash_insitu/{trace,<COUNTER>}/**.csv -> a markdown table.
The big leaf launches (2^21 leaves of 256 columns) come in dispatch order: 9 inside proofs (3 warm), then 9 isolated (3 warm)."""
import collections
import csv
import glob
import os
import sys

P = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/hash_insitu"
KEY = "hash_rows_vec_kernel"


def rows(d, suffix):
    out = []
    for f in glob.glob(os.path.join(P, d, "**", "*" + suffix), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


tr = [r for r in rows("trace", "kernel_trace.csv") if KEY in r["Kernel_Name"]]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
grid = [int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0) for r in tr]
big = max(grid) if grid else 0
sel = [i for i, g in enumerate(grid) if g == big]
print("# leaf hashing (`zk::hash_rows_vec_kernel`, 2^21 leaves x 256 columns) inside a proof and alone\n")
print("dispatches of that size in the kernel trace: %d (expected 18: 9 in proofs, 9 isolated; the first 3 of each warm)\n" % len(sel))
cols = collections.OrderedDict()
cols["duration_us"] = [dur[i] for i in sel]
for d in sorted(os.listdir(P)):
    if d == "trace" or not os.path.isdir(os.path.join(P, d)):
        continue
    rr = [r for r in rows(d, "counter_collection.csv") if KEY in r["Kernel_Name"]]
    by = collections.OrderedDict()
    for r in rr:
        by.setdefault(int(r["Dispatch_Id"]), {"g": int(r.get("Grid_Size", "0") or 0), "v": 0.0})
        by[int(r["Dispatch_Id"])]["v"] += float(r["Counter_Value"])
    gmax = max((v["g"] for v in by.values()), default=0)
    vals = [v["v"] for k, v in sorted(by.items()) if v["g"] == gmax]
    cols[d] = vals
names = list(cols)
print("| # | where | " + " | ".join(names) + " |")
print("|---|---|" + "---|" * len(names))
n = max(len(v) for v in cols.values())
for i in range(n):
    where = "in proof" if i < 9 else "isolated"
    if i % 9 < 3:
        where += " (warm)"
    print("| %d | %s | " % (i, where) + " | ".join(("%.6g" % cols[c][i]) if i < len(cols[c]) else "-" for c in names) + " |")
print()
for c in names:
    v = cols[c]
    if len(v) >= 18:
        a, b = sum(v[3:9]) / 6, sum(v[12:18]) / 6
        print("* %s: in proof %.6g, isolated %.6g, ratio %.4f" % (c, a, b, a / b if b else float("nan")))
