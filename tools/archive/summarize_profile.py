#!/usr/bin/env python3
"""Copy the rocprofv3 outputs worth keeping from gpurun_out/prof into profiles/ and
derive per-launch HBM traffic of the NTT pass kernel from the PMC passes.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of the bytes of
coalesced streaming reads (MI355X_MICROARCH.md, section HBM) -> doubled here."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (tools/archive/ -> the repository root)
src = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

kt = sys.argv[2] if len(sys.argv) > 2 else "kt"
stats = glob.glob(os.path.join(src, kt, "**", "*_kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
    cmdline = open(os.path.join(src, kt + "_cmd.txt")).read().strip() if os.path.exists(os.path.join(src, kt + "_cmd.txt")) else "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(dst, tag + "_kernel_stats.md"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- %s (%s)\n\n" % (cmdline, tag))
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows:
            f.write("| %s | %s | %.3f | %.2f | %s |\n" % (r["Name"].split("(")[0], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                         float(r["AverageNs"]) / 1e3, r["Percentage"]))
log = os.path.join(src, "bench_under_prof.log" if kt == "kt" else "bench_under_prof_s1.log")
if os.path.exists(log):
    for line in open(log):
        if line.startswith("{"):
            open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w").write(line)

def pmc(name, counter):
    files = glob.glob(os.path.join(src, name, "**", "*_counter_collection.csv"), recursive=True)
    vals = []
    if files:
        for r in csv.DictReader(open(files[0])):
            if "ntt_pass_kernel<4" in r["Kernel_Name"] and ", 2, 5, " in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
        shutil.copy(files[0], os.path.join(dst, "%s_%s_counter_collection.csv" % (tag, name)))
    return vals

fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
if fetch and write:
    f_avg = sum(fetch) / len(fetch) * 1024.0 * 2.0     # KiB -> B, gfx950 x2 correction
    w_avg = sum(write) / len(write) * 1024.0
    out = {"kernel": "zk::ntt_pass_kernel<4,*,2,5,*> (the 1024-row two-column passes of a 2^20 x 256 LDE, in-proof and replayed by tools/profile_ntt.py)",
           "workload": "2^20 x 256, mean over all such launches of the run", "measured_at": tag + (" @ " + sys.argv[3] if len(sys.argv) > 3 else ""),
           "fetch_size_kib_raw_mean": sum(fetch) / len(fetch), "write_size_kib_mean": sum(write) / len(write),
           "fetch_correction": "x2 (gfx950 FETCH_SIZE half-count)", "hbm_bytes_per_launch": f_avg + w_avg,
           "algorithmic_bytes_per_launch": 8.0 * (1 << 28)}
    json.dump(out, open(os.path.join(dst, "pmc_ntt_pass.json"), "w"), indent=1)
    print(out)
