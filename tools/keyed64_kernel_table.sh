#!/bin/bash
# The kernel table of the 64-proof machine-mode compression (three calls of zkhip_prove_machine_verifier over 64 keyed transcript proofs, and the lock-step batch that makes them):
# rocprofv3 --kernel-trace --stats over tools/join_breakdown.py --keyed 64 -> gpurun_out/prof6/keyed64_kernel_stats.md (copied to profiles/r06_keyed64_kernel_stats.md)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/prof6; mkdir -p $P; rm -rf $P/k64
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/k64 -o run -- python3 tools/join_breakdown.py --keyed 64 > $P/k64.log 2>&1
F=$(find $P/k64 -name "*kernel_stats.csv" | head -1)
python3 - "$F" > $P/keyed64_kernel_stats.md <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("# rocprofv3 --kernel-trace --stats -- python3 tools/join_breakdown.py --keyed 64\n")
print("Three calls of `zkhip_prove_machine_verifier` (64 keyed transcript proofs -> ONE proof each) and the lock-step batches that make the inner proofs (`*_batch` kernels). Per compression: divide the un-batched kernels by three.\n")
print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for r in rows[:40]:
    nm = r["Name"][:r["Name"].rfind("(")] if r["Name"].endswith(")") else r["Name"]
    print("| %s | %s | %.2f | %.1f | %s |" % (nm, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
head -30 $P/keyed64_kernel_stats.md
find $P -name "*kernel_trace.csv" -size +20M -delete
