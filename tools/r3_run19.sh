cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python3 bench.py > gpurun_out/r3_bench2.json 2> gpurun_out/r3_bench2.err; echo "bench rc=$?"; tail -3 gpurun_out/r3_bench2.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench2.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","transcripts_per_s")}); print(d["batch64"]); print(d["roofline"]["frac"], d["cpu_baseline"])
PY
rm -rf gpurun_out/r3_b64; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_b64 -o run -- python3 tools/lockstep_trace.py 16 6 > gpurun_out/r3_b64.log 2>&1
tail -1 gpurun_out/r3_b64.log
python3 tools/window_stats.py $(find gpurun_out/r3_b64 -name "*kernel_trace.csv") 72 > gpurun_out/r3_b64_window.txt; head -12 gpurun_out/r3_b64_window.txt
python3 tools/gap_report.py $(find gpurun_out/r3_b64 -name "*kernel_trace.csv") 72 > gpurun_out/r3_b64_gaps.txt; head -3 gpurun_out/r3_b64_gaps.txt
find gpurun_out/r3_b64 -name "*kernel_trace.csv" -delete
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5
