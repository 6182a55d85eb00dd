cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/r3_wide
timeout 1500 python3 -m pytest tests/test_gpu_sha256_chip.py tests/test_gpu_air.py tests/test_gpu_p2chip.py tests/test_gpu_keyed_machine.py tests/test_gpu_chips_air.py tests/test_gpu_machine.py tests/test_gpu_fri_chip.py tests/test_gpu_lockstep.py -m gpu -x -q 2>&1 | tail -4
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_wide -o run -- python3 tools/airq_fixed.py > gpurun_out/r3_wide.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r3_wide/**/*kernel_stats.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows:
    if "quotient" in r["Name"]: print("%6d calls %9.1f us avg  %s" % (int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:100]))
PY
find gpurun_out/r3_wide -name "*kernel_trace.csv" -delete
