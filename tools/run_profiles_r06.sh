#!/bin/bash
# The rocprofv3 runs behind profiles/r06_* (run on the GPU box through gpurun; outputs under gpurun_out/prof6).
# Kernel trace + stats and the PMC counters are SEPARATE runs (counters only, no trace domains).  Every attempt is counted in $P/attempts.txt.
# usage: run_profiles_r06.sh [hash|contract|all]
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
find $P -name "*kernel_trace.csv" -size +40M -delete
find $P -name "*counter_collection.csv" -size +8M -delete
du -sh $P
cat $P/attempts.txt
