#!/usr/bin/env python3
"""Turn the outputs of tools/run_profiles_r06.sh evidence (gpurun_out/prof6) into the tracked r06 evidence: kernel tables (contract command; ONE shard at a
time, per proof), the HBM traffic per launch of the LDE kernels from the FETCH_SIZE / WRITE_SIZE passes (KiB; on gfx950 FETCH_SIZE counts half of the bytes
of coalesced streaming reads -- MI355X_MICROARCH.md, section HBM -- so it is doubled), profiles/pmc_ntt_pass.json for bench.py's `roofline.traffic`.
usage: summarize_profile_r06.py <prof dir> [commit]   (run on the GPU box at the end of the evidence pass, and again here to copy into profiles/)"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof6")
commit = sys.argv[2] if len(sys.argv) > 2 else ""
dst = os.path.join(ROOT, "profiles")
PROOFS_IN_SINGLE = 6


def stats_table(name, tag, per=None):
    files = glob.glob(os.path.join(src, name, "**", "*_kernel_stats.csv"), recursive=True)
    if not files:
        return
    shutil.copy(files[0], os.path.join(dst, tag + "_kernel_stats.csv"))
    cmd = open(os.path.join(src, name + "_cmd.txt")).read().strip() if os.path.exists(os.path.join(src, name + "_cmd.txt")) else name
    rows = list(csv.DictReader(open(files[0])))
    with open(os.path.join(dst, tag + "_kernel_stats.md"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- %s (%s%s)\n\n" % (cmd, tag, " @ " + commit if commit else ""))
        if per:
            f.write("%d proofs, one in flight, nothing else in the process: `per proof` = total / %d.\n\n" % (per, per))
            f.write("| kernel | calls per proof | ms per proof | avg us | % |\n|---|---|---|---|---|\n")
        else:
            f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows:
            nm = r["Name"][:r["Name"].rfind("(")] if r["Name"].endswith(")") else r["Name"]
            if per:
                f.write("| %s | %.1f | %.3f | %.2f | %s |\n" % (nm, float(r["Calls"]) / per, float(r["TotalDurationNs"]) / 1e6 / per, float(r["AverageNs"]) / 1e3, r["Percentage"]))
            else:
                f.write("| %s | %s | %.3f | %.2f | %s |\n" % (nm, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
    log = os.path.join(src, name + ".log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("{"):
                open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w").write(line)


stats_table("contract", "r06_contract")
stats_table("single", "r06_single_shard", per=PROOFS_IN_SINGLE)
for f in ("compress64_phases.txt", "keyed64_phases.txt", "tree_phases.txt", "compress64_phases_hostwalk.txt", "keyed64_phases_hostwalk.txt", "tree_phases_hostwalk.txt",
          "bench_default.json", "bench_driver_command.json", "hash_dispatches.md"):
    if os.path.exists(os.path.join(src, f)) and os.path.getsize(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, "r06_" + f))


def pmc(name, counter, match):
    files = glob.glob(os.path.join(src, name, "**", "*_counter_collection.csv"), recursive=True)
    vals = []
    if files:
        for r in csv.DictReader(open(files[0])):
            if match(r["Kernel_Name"]) and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return vals


out = {"measured_at": "r06" + (" @ " + commit if commit else ""), "fetch_correction": "x2 (gfx950 FETCH_SIZE half-count)", "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 tools/single_shard_trace.py (six headline proofs, one in flight)", "kernels": {}}
for label, match, alg in (("zk::ntt_pass_kernel<4,*,2,5,*> (1024-row two-column passes of a 2^20 x 256 LDE)", lambda k: "ntt_pass_kernel<4" in k and ", 2, 5, " in k, 8.0 * (1 << 28)),
                          ("zk::lde_fused_kernel (second inverse pass + first forward pass of both cosets)", lambda k: "lde_fused_kernel" in k, 12.0 * (1 << 28))):
    fetch, write = pmc("pmc_FETCH_SIZE", "FETCH_SIZE", match), pmc("pmc_WRITE_SIZE", "WRITE_SIZE", match)
    if fetch and write:
        f_avg = sum(fetch) / len(fetch) * 1024.0 * 2.0
        w_avg = sum(write) / len(write) * 1024.0
        out["kernels"][label] = {"launches": len(fetch), "fetch_size_kib_raw_mean": sum(fetch) / len(fetch), "write_size_kib_mean": sum(write) / len(write),
                                 "hbm_bytes_per_launch": f_avg + w_avg, "algorithmic_bytes_per_launch": alg, "ratio": (f_avg + w_avg) / alg}
if out["kernels"]:
    first = list(out["kernels"].values())[0]
    out.update({"kernel": list(out["kernels"].keys())[0], "workload": "2^20 x 256, mean over all such launches of the run",
                "hbm_bytes_per_launch": first["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": first["algorithmic_bytes_per_launch"]})
    json.dump(out, open(os.path.join(src, "pmc_ntt_pass.json"), "w"), indent=1)
    shutil.copy(os.path.join(src, "pmc_ntt_pass.json"), os.path.join(dst, "pmc_ntt_pass.json"))
    print(json.dumps(out, indent=1))
else:
    print("no counter rows for the LDE kernels under", src)
