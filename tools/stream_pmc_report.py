#!/usr/bin/env python3
"""PMC passes of tools/single_shard_trace.py (tools/run_profiles_r05.sh: <dir>/stream_<COUNTER>/): mean per launch for the streaming passes
over the finished LDE of a headline proof (quotient_kernel<4>, rowdot_regs_kernel<4>), the leaf hash and the opening kernel -> a markdown
table.  Only launches of the headline shape enter (the kernel's largest launches: counters of the small quotient-chunk launches are
dropped by the `big` filter: a launch counts when its FETCH-independent grid size is the kernel's maximum)."""
import collections
import csv
import glob
import os
import sys

P = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else "stream_"
want = collections.OrderedDict([
    ("quotient_kernel<4>", lambda k: "quotient_kernel<4>" in k and "batch" not in k),
    ("rowdot_regs_kernel<4>", lambda k: "rowdot_regs_kernel<4>" in k and "batch" not in k),
    ("hash_rows_vec_kernel", lambda k: "hash_rows_vec_kernel" in k and "batch" not in k),
    ("open_partial_kernel", lambda k: "open_partial" in k and "batch" not in k),
    ("ntt_pass I1", lambda k: "ntt_pass_kernel<4, true, 2, 5, 2>" in k),
    ("ntt_pass F2", lambda k: "ntt_pass_kernel<4, false, 2, 5, 1>" in k),
])
tab = collections.OrderedDict((n, {}) for n in want)
for d in sorted(os.listdir(P)):
    if not d.startswith(prefix) or not os.path.isdir(os.path.join(P, d)):
        continue
    c = d[len(prefix):]
    for f in glob.glob(os.path.join(P, d, "**", "*counter_collection.csv"), recursive=True):
        by = collections.defaultdict(float)
        kn, grid = {}, {}
        for r in csv.DictReader(open(f)):
            by[r["Dispatch_Id"]] += float(r["Counter_Value"])
            kn[r["Dispatch_Id"]] = r["Kernel_Name"]
            grid[r["Dispatch_Id"]] = int(r.get("Grid_Size", 0) or 0)
        for n, m in want.items():
            ids = [i for i in by if m(kn[i])]
            if not ids:
                continue
            gmax = max(grid[i] for i in ids)
            ids = [i for i in ids if grid[i] == gmax]
            tab[n][c] = (sum(by[i] for i in ids) / len(ids), len(ids))
cols = sorted({c for v in tab.values() for c in v})
print("# PMC counters, mean per launch of the headline-shape launches (one counter per pass; `tools/single_shard_trace.py`: six 2^20 x 256 proofs, one in flight)\n")
print("| counter | " + " | ".join(tab) + " |")
print("|---|" + "---|" * len(tab))
for c in cols:
    print("| %s | " % c + " | ".join(("%.5g (%d)" % tab[n][c]) if c in tab[n] else "-" for n in tab) + " |")
