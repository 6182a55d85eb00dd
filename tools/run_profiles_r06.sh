#!/bin/bash
# The rocprofv3 runs behind profiles/r06_* (run on the GPU box through gpurun; outputs under gpurun_out/prof6).
# Kernel trace + stats and the PMC counters are SEPARATE runs (counters only, no trace domains).  Every attempt is counted in $P/attempts.txt.
# usage: run_profiles_r06.sh [hash|evidence|all]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/prof6
mkdir -p $P
WHAT=${1:-all}
CONTRACT="python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-multichip --no-execution --no-batch64 --no-recursion16"
run_kt() {   # name, command...
    local name=$1; shift
    echo "$*" > $P/${name}_cmd.txt
    for attempt in 1 2 3 4; do
        rm -rf $P/$name
        if rocprofv3 --kernel-trace --stats --output-format csv -d $P/$name -o run -- "$@" > $P/${name}.log 2>&1; then
            echo "$name: attempt $attempt survived" >> $P/attempts.txt; break
        fi
        echo "$name: attempt $attempt died" >> $P/attempts.txt
    done
}
if [ "$WHAT" = hash ] || [ "$WHAT" = all ]; then
# 1. the headline proofs alone, four in flight: every 2^21 x 256 leaf-hash dispatch with its duration, and (own pass) its GRBM_GUI_ACTIVE
run_kt hash4 $CONTRACT
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $P/hash4_grbm -o run -- $CONTRACT > $P/hash4_grbm.log 2>&1 || echo "pmc GRBM died" >> $P/attempts.txt
python3 tools/hash_dispatch_report.py $P/hash4 $P/hash4_grbm "the contract command's proofs (four in flight)" > $P/hash_dispatches.md 2>&1
fi
if [ "$WHAT" = evidence ] || [ "$WHAT" = all ]; then
# 2. the contract command (the driver's: default line) under the kernel trace
run_kt contract python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-multichip --no-execution
# 3. ONE headline shard at a time, proofs only (tools/single_shard_trace.py: six proofs): the per-proof kernel table, and the HBM counters of the LDE passes
run_kt single python3 tools/single_shard_trace.py
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $P/pmc_$c -o run -- python3 tools/single_shard_trace.py > $P/pmc_$c.log 2>&1 || echo "pmc $c died" >> $P/attempts.txt
done
# 4. the phases of the two compressions and of the tree's top (A/B build, no profiler; ZKHIP_REC_HOST=1: the host's walk of round 5 beside the device's)
python3 tools/join_breakdown.py --sha 64 > $P/compress64_phases.log 2>&1; tail -20 $P/compress64_phases.log > $P/compress64_phases.txt
python3 tools/join_breakdown.py --keyed 64 > $P/keyed64_phases.log 2>&1; tail -22 $P/keyed64_phases.log > $P/keyed64_phases.txt
python3 tools/tree_breakdown.py 4 > $P/tree_phases.log 2>&1; grep -E "machine verifier|top over|chips prover" $P/tree_phases.log | tail -22 > $P/tree_phases.txt
ZKHIP_REC_HOST=1 python3 tools/join_breakdown.py --sha 64 2>&1 | grep -E "shard verifier\]|compress" | tail -9 > $P/compress64_phases_hostwalk.txt
ZKHIP_REC_HOST=1 python3 tools/join_breakdown.py --keyed 64 2>&1 | grep -E "machine verifier\]|compress" | tail -6 > $P/keyed64_phases_hostwalk.txt
ZKHIP_REC_HOST=1 python3 tools/tree_breakdown.py 4 2>&1 | grep -E "machine verifier\]|top over" | tail -6 > $P/tree_phases_hostwalk.txt
# 5. the default line and the driver's command, no profiler
python3 bench.py > $P/bench_default.json 2> $P/bench_default.err
python3 bench.py --steps 20 --warmup 5 > $P/bench_driver_command.json 2> $P/bench_driver_command.err
python3 tools/summarize_profile_r06.py $P > $P/summary.log 2>&1; tail -30 $P/summary.log
fi
find $P -name "*kernel_trace.csv" -size +40M -delete
find $P -name "*counter_collection.csv" -size +8M -delete
du -sh $P
cat $P/attempts.txt
