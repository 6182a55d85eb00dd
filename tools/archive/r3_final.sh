cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python3 bench.py > gpurun_out/r3_bench3.json 2> gpurun_out/r3_bench3.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench3.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","transcripts_per_s","single_shard_latency_ms")}); print(d["batch64"]["ms"], d["batch64"]["ms_with_verify_inside"], d["batch64"]["one_stream_per_worker_ms"], d["batch64"]["same_bytes_both_ways"]); print(d["roofline"]["frac"], d["roofline"]["lde"]["ms"], d["cpu_baseline"]["value"]); print(d["recursion16"])
PY
