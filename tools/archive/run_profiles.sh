#!/bin/bash
# The rocprofv3 runs behind profiles/ (run on the GPU box through gpurun; outputs under gpurun_out/prof).
# Kernel trace + stats and the PMC counters are SEPARATE runs (counters only, no trace domains).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/prof
mkdir -p $P
echo "python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline" > $P/kt_cmd.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -o run -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $P/bench_under_prof.log 2>&1
echo "python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --streams 1" > $P/kt1_cmd.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt1 -o run -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --streams 1 > $P/bench_under_prof_s1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -o run -- python3 tools/profile_ntt.py > $P/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -o run -- python3 tools/profile_ntt.py > $P/pmc_write.log 2>&1
find $P -name "*.csv" | head -20
