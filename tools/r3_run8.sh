cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3_q1
timeout 1500 python -m pytest tests/test_gpu_air.py tests/test_gpu_chips_air.py -m gpu -x -q 2>&1 | tail -8
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_q1/kt -o run -- python3 tools/airq_fixed.py > gpurun_out/r3_q1/airq.log 2>&1
cat gpurun_out/r3_q1/airq.log
find gpurun_out/r3_q1 -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r3_q1/kt/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3)
PY
