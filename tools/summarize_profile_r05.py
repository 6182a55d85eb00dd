#!/usr/bin/env python3
"""Copy the rocprofv3 outputs of tools/run_profiles_r05.sh (gpurun_out/prof5) into profiles/ as r05_*: the kernel tables of the multi-chip
shard, of the headline shard alone (one in flight) and of the contract command, the multi-chip phase table, the counter table of the
streaming passes, the default bench line and the attempt counts of every profiled command.
    python3 tools/summarize_profile_r05.py [commit [name ...]]     (names: multichip kt1 kt compress64 tree -- only those runs are copied; none: all that exist)"""
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof5")
dst = os.path.join(ROOT, "profiles")
commit = sys.argv[1] if len(sys.argv) > 1 else ""
only = set(sys.argv[2:])
attempts = open(os.path.join(src, "attempts.txt")).read().strip().split("\n") if os.path.exists(os.path.join(src, "attempts.txt")) else []

for name, tag in (("multichip", "multichip"), ("kt1", "streams1"), ("kt", "contract"), ("compress64", "compress64"), ("tree", "tree")):
    stats = glob.glob(os.path.join(src, name, "**", "*_kernel_stats.csv"), recursive=True)
    if not stats or (only and name not in only):
        continue
    tag = "r05_" + tag
    shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
    cmd = open(os.path.join(src, name + "_cmd.txt")).read().strip()
    rows = list(csv.DictReader(open(stats[0])))
    tries = [a for a in attempts if a.startswith(name + ":")]
    with open(os.path.join(dst, tag + "_kernel_stats.md"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- %s (%s%s)\n\n" % (cmd, tag, " @ " + commit if commit else ""))
        f.write("(attempts of this command under the profiler: %s)\n\n" % ("; ".join(t.split(": ", 1)[1] for t in tries) or "one"))
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows:
            f.write("| %s | %s | %.3f | %.2f | %s |\n" % ((r["Name"][:r["Name"].rfind("(")] if r["Name"].endswith(")") else r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                         float(r["AverageNs"]) / 1e3, r["Percentage"]))
    times = [line for line in open(os.path.join(src, name + ".log")) if "transcripts:" in line]
    if times:
        open(os.path.join(dst, tag + "_times_under_rocprof.txt"), "w").write("".join(times))
    for line in open(os.path.join(src, name + ".log")):
        if line.startswith("{"):
            open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w").write(line)
for a, b in (("stream_pmc.md", "r05_stream_pass_pmc.md"), ("bench_default.json", "r05_bench_default.json"), ("multichip_phases.txt", "r05_multichip_phases_under_rocprof.txt"),
             ("multichip_phases_plain.txt", "r05_multichip_phases.txt"), ("compress64_phases.txt", "r05_compress64_phases.txt"), ("tree_phases.txt", "r05_tree_phases.txt")):
    if os.path.exists(os.path.join(src, a)) and not only:
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))
print("\n".join(attempts))
