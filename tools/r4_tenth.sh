cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4k; mkdir -p $O
timeout 300 python3 tools/multichip_breakdown.py > $O/mc.log 2>&1; tail -24 $O/mc.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 $GRAFT_REPO_ROOT/tools/multichip_breakdown.py) > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time total %.1f ms over 4 proofs + setup" % (tot / 1e6))
for r in rows[:26]:
    print("%-80s calls %6s total %9.2f ms avg %9.1f us" % (r["Name"][:80], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
find $O -name "*.csv" -size +3000k -delete
