#!/bin/bash
# The validation of the final tree of round 6 on a GPU box (through gpurun): the GPU suite with its ten slowest tests, smoke(), the driver's bench
# command, the default line, and the phase tables of the recursion machines (A/B build, no profiler).  Outputs under gpurun_out/prof6.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
P=gpurun_out/prof6; mkdir -p $P
timeout 1500 python3 -m pytest tests -q -m gpu -x --durations=10 > $P/gpu_suite_final.txt 2>&1; echo "pytest rc $?"; tail -16 $P/gpu_suite_final.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $P/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $P/smoke.log
timeout 600 python3 bench.py --steps 20 --warmup 5 > $P/bench_driver_command.json 2> $P/bench_driver_command.err; echo "bench rc $?"
timeout 600 python3 bench.py > $P/bench_default.json 2> $P/bench_default.err; echo "bench default rc $?"
timeout 300 python3 tools/join_breakdown.py --sha 64 > $P/compress64_phases.log 2>&1; tail -20 $P/compress64_phases.log > $P/compress64_phases.txt
timeout 300 python3 tools/join_breakdown.py --keyed 64 > $P/keyed64_phases.log 2>&1; tail -22 $P/keyed64_phases.log > $P/keyed64_phases.txt
timeout 300 python3 tools/tree_breakdown.py 4 > $P/tree_phases.log 2>&1; grep -E "machine verifier|top over|chips prover" $P/tree_phases.log | tail -22 > $P/tree_phases.txt
cat $P/tree_phases.txt
python3 - <<'PY'
import json
for f in ("bench_driver_command", "bench_default"):
    d = json.loads(open("gpurun_out/prof6/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["roofline"]["frac"], d["batch64"]["ms"], d["recursion"]["tree_ms"], d["multichip"]["ms_per_shard"], d["multichip"]["compressed_ms"], d["execution22"]["core_plus_compress_ms"], d["cpu_baseline"]["value"])
PY
