#!/usr/bin/env python3
"""Copy the rocprofv3 outputs of tools/run_profiles_r03.sh (gpurun_out/prof3) into profiles/ as r03_* and derive the HBM traffic
per launch of the LDE kernels from the PMC passes.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of the
bytes of coalesced streaming reads (MI355X_MICROARCH.md, section HBM) -> doubled here."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (tools/archive/ -> the repository root)
src = os.path.join(ROOT, "gpurun_out", "prof3")
dst = os.path.join(ROOT, "profiles")
commit = sys.argv[1] if len(sys.argv) > 1 else ""

for name in ("kt", "kt1", "r0", "big21", "big22"):
    stats = glob.glob(os.path.join(src, name, "**", "*_kernel_stats.csv"), recursive=True)
    if not stats:
        continue
    tag = "r03_" + {"kt": "contract", "kt1": "streams1", "r0": "r0shape", "big21": "2pow21", "big22": "2pow22"}[name]
    shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
    cmd = open(os.path.join(src, name + "_cmd.txt")).read().strip()
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(dst, tag + "_kernel_stats.md"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- %s (%s%s)\n\n" % (cmd, tag, " @ " + commit if commit else ""))
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows:
            f.write("| %s | %s | %.3f | %.2f | %s |\n" % ((r["Name"][:r["Name"].rfind("(")] if r["Name"].endswith(")") else r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                         float(r["AverageNs"]) / 1e3, r["Percentage"]))
    for line in open(os.path.join(src, name + ".log")):
        if line.startswith("{"):
            open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w").write(line)
if os.path.exists(os.path.join(src, "bench_default.json")):
    shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, "r03_bench_default.json"))


def pmc(name, counter, match):
    files = glob.glob(os.path.join(src, name, "**", "*_counter_collection.csv"), recursive=True)
    vals = []
    if files:
        for r in csv.DictReader(open(files[0])):
            if match(r["Kernel_Name"]) and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
        shutil.copy(files[0], os.path.join(dst, "r03_%s_counter_collection.csv" % name))
    return vals


out = {"measured_at": "r03" + (" @ " + commit if commit else ""), "fetch_correction": "x2 (gfx950 FETCH_SIZE half-count)", "kernels": {}}
for label, match, alg in (("zk::ntt_pass_kernel<4,*,2,5,*> (1024-row two-column passes of a 2^20 x 256 LDE)", lambda k: "ntt_pass_kernel<4" in k and ", 2, 5, " in k, 8.0 * (1 << 28)),
                          ("zk::lde_fused_kernel (second inverse pass + first forward pass of both cosets)", lambda k: "lde_fused_kernel" in k, 12.0 * (1 << 28))):
    fetch, write = pmc("pmc_fetch", "FETCH_SIZE", match), pmc("pmc_write", "WRITE_SIZE", match)
    if fetch and write:
        f_avg = sum(fetch) / len(fetch) * 1024.0 * 2.0
        w_avg = sum(write) / len(write) * 1024.0
        out["kernels"][label] = {"launches": len(fetch), "fetch_size_kib_raw_mean": sum(fetch) / len(fetch), "write_size_kib_mean": sum(write) / len(write),
                                 "hbm_bytes_per_launch": f_avg + w_avg, "algorithmic_bytes_per_launch": alg, "ratio": (f_avg + w_avg) / alg}
if out["kernels"]:
    first = list(out["kernels"].values())[0]
    out.update({"kernel": list(out["kernels"].keys())[0], "workload": "2^20 x 256, mean over all such launches of the run",
                "hbm_bytes_per_launch": first["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": first["algorithmic_bytes_per_launch"]})
    json.dump(out, open(os.path.join(dst, "pmc_ntt_pass.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
