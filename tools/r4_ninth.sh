cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4j; mkdir -p $O
timeout 300 python3 tools/join_breakdown.py 1 > $O/join1.log 2>&1; tail -20 $O/join1.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/prof -o run -- python3 $GRAFT_REPO_ROOT/tools/join_breakdown.py 1) > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last join call: find the last p2r_rows_kernel and take everything from the fold trace kernel before it to the end
idx = [i for i, r in enumerate(rows) if "p2r_rows_kernel" in r["Kernel_Name"]]
i0 = idx[-1]
seg = rows[i0:]
t0 = int(seg[0]["Start_Timestamp"]); t1 = int(seg[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print("last join: %d kernels from the P2R rows kernel on, span %.2f ms, kernel time %.2f ms" % (len(seg), (t1 - t0) / 1e6, busy / 1e6))
import collections
c = collections.Counter(); d = collections.Counter()
for r in seg:
    n = r["Kernel_Name"].split("(")[0][:60]; c[n] += 1; d[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, k in c.most_common(30): print("  %-62s %4d launches %8.1f us" % (n, k, d[n] / 1e3))
# gaps
gaps = sorted(((int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"])) / 1e3, seg[i]["Kernel_Name"].split("(")[0][:40], seg[i + 1]["Kernel_Name"].split("(")[0][:40]) for i in range(len(seg) - 1))
print("largest gaps (us):")
for g in gaps[-25:]: print("   %8.1f  %s -> %s" % g)
print("sum of gaps > 20 us: %.2f ms; sum of gaps <= 20 us: %.2f ms" % (sum(g[0] for g in gaps if g[0] > 20) / 1e3, sum(g[0] for g in gaps if 0 < g[0] <= 20) / 1e3))
PY
find $O -name "*.csv" -size +3000k -delete
