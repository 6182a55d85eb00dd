cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/r3_q2; mkdir -p gpurun_out/r3_q2
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3_q2/kt -o run -- python3 tools/airq_fixed.py > gpurun_out/r3_q2/airq.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r3_q2/kt/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "quotient_air" in r["Kernel_Name"]]
for r in rows:
    print(r["Kernel_Name"][9:60], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, "us  grid", r.get("Grid_Size"), "lds", r.get("LDS_Block_Size"), "vgpr", r.get("VGPR_Count"))
PY
find gpurun_out/r3_q2 -name "*kernel_trace.csv" -delete
