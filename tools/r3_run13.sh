cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/r3_lst
for cfg in "64 1" "32 2" "16 4" "0 1"; do timeout 200 python3 tools/lockstep_trace.py $cfg | tail -1; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_lst -o run -- python3 tools/lockstep_trace.py 64 1 > gpurun_out/r3_lst.log 2>&1
tail -1 gpurun_out/r3_lst.log
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r3_lst/**/*kernel_stats.csv",recursive=True)
if f:
    rows=list(csv.DictReader(open(f[0])))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print("kernels total %.1f ms over the whole run (3 batches + setup)" % (tot/1e6))
    for r in rows[:30]: print("%8d %9.1f us avg %8.2f ms total  %s" % (int(r["Calls"]), float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Name"][:90]))
PY
find gpurun_out/r3_lst -name "*kernel_trace.csv" -delete
timeout 600 python3 tools/lockstep_time.py 64 > gpurun_out/r3_ls64.log 2>&1; echo "rc64=$?"; tail -25 gpurun_out/r3_ls64.log
