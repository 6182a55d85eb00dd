cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/single; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o run -- python3 $GRAFT_REPO_ROOT/tools/single_shard_trace.py) > $O/prof.log 2>&1
tail -7 $O/prof.log
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last proof: from the last tagged I1-like first LDE kernel... take the last 17.5 ms
t_end = int(rows[-1]["End_Timestamp"])
seg = [r for r in rows if int(r["Start_Timestamp"]) >= t_end - 17.0e6]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print("last 17 ms: %d kernels, kernel time %.2f ms" % (len(seg), busy / 1e6))
gaps = []
for i in range(len(seg) - 1):
    g = (int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"])) / 1e3
    if g > 15: gaps.append((g, (int(seg[i]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6, seg[i]["Kernel_Name"].split("(")[0][-40:], seg[i + 1]["Kernel_Name"].split("(")[0][-40:]))
print("gaps > 15 us: %d, total %.2f ms" % (len(gaps), sum(g[0] for g in gaps) / 1e3))
for g in sorted(gaps, reverse=True)[:30]: print("  %8.1f us at %6.2f ms  %s -> %s" % g)
c = collections.Counter(); d = collections.Counter()
for r in seg:
    n = r["Kernel_Name"].split("(")[0][-50:]; c[n] += 1; d[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, t in d.most_common(22): print("  %-52s %4d launches %9.1f us" % (n, c[n], t / 1e3))
PY
rm -rf $O/prof
