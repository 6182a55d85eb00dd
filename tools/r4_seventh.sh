cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4h; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_big.py tests/test_gpu_recursion.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 300 python3 tools/big_lde_time.py > $O/big_lde.log 2>&1; cat $O/big_lde.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_big -o run -- python3 $GRAFT_REPO_ROOT/tools/big_lde_time.py 21x256) > $O/prof_big.log 2>&1
f=$(find $O/prof_big -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:6]:
    print("%-90s calls %6s total %9.2f ms avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
find $O -name "*.csv" -size +3000k -delete
timeout 300 python3 tools/join_breakdown.py 16 > $O/join.log 2>&1; tail -22 $O/join.log
