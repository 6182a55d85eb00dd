cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4d; mkdir -p $O
for try in 1 2 3; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 $GRAFT_REPO_ROOT/tools/join_breakdown.py 16) > $O/prof.log 2>&1 && break
done
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:28]:
    print("%-110s calls %6s total %9.2f ms avg %9.1f us" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
find $O/prof -name "*.csv" -size +3000k -delete
