cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for abl in 0 2 4; do
rm -rf gpurun_out/r3_abl
ZKHIP_AIRQ_ABL=$abl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_abl -o run -- python3 tools/airq_ablate.py > gpurun_out/r3_abl.log 2>&1
echo "ABL=$abl"; python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r3_abl/**/*kernel_stats.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows:
    if "wide" in r["Name"]: print("%6d calls %9.1f us avg  %s" % (int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:70]))
PY
done
rm -rf gpurun_out/r3_abl
