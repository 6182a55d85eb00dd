#!/usr/bin/env python3
"""PMC passes of tools/profile_fused.py (tools/run_profiles_r04.sh: <dir>/pass_<COUNTER>/): mean per launch for the LDE's pass kernels and the
fused launch -> a markdown table (I1 = ntt_pass_kernel<4,true,2,5,4>, F2 = <4,false,2,5,3>: the names zkhip_ntt_pass launches them under)."""
import collections
import csv
import glob
import os
import sys

P = sys.argv[1]
names = {"I1 (inverse, strided 128-B chunks in -> one block out)": lambda k: "ntt_pass_kernel<4, true, 2, 5, 4>" in k,
         "F2 (forward, contiguous, in place)": lambda k: "ntt_pass_kernel<4, false, 2, 5, 3>" in k,
         "fused middle launch": lambda k: "lde_fused_kernel" in k}
tab = collections.OrderedDict((n, {}) for n in names)
for d in sorted(os.listdir(P)):
    if not d.startswith("pass_") or not os.path.isdir(os.path.join(P, d)):
        continue
    c = d[5:]
    acc = {n: [0.0, 0] for n in names}
    for f in glob.glob(os.path.join(P, d, "**", "*counter_collection.csv"), recursive=True):
        by = collections.defaultdict(float)
        kn = {}
        for r in csv.DictReader(open(f)):
            by[r["Dispatch_Id"]] += float(r["Counter_Value"])
            kn[r["Dispatch_Id"]] = r["Kernel_Name"]
        for i, v in by.items():
            for n, m in names.items():
                if m(kn[i]):
                    acc[n][0] += v
                    acc[n][1] += 1
    for n in names:
        if acc[n][1]:
            tab[n][c] = acc[n][0] / acc[n][1]
cols = sorted({c for v in tab.values() for c in v})
print("# PMC counters of the LDE's launches at 2^20 x 256, mean per launch (one counter per pass; `tools/profile_fused.py`)\n")
print("| counter | " + " | ".join(tab) + " | I1 / F2 |")
print("|---|" + "---|" * (len(tab) + 1))
ks = list(tab)
for c in cols:
    vals = [tab[n].get(c) for n in ks]
    ratio = (vals[0] / vals[1]) if vals[0] is not None and vals[1] else float("nan")
    print("| %s | " % c + " | ".join("%.4g" % v if v is not None else "-" for v in vals) + " | %.3f |" % ratio)
