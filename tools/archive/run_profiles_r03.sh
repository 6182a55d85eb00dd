#!/bin/bash
# The rocprofv3 runs behind profiles/r03_* (run on the GPU box through gpurun; outputs under gpurun_out/prof3).
# Kernel trace + stats and the PMC counters are SEPARATE runs (counters only, no trace domains).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/prof3
rm -rf $P; mkdir -p $P
run_kt() {   # name, command...
    local name=$1; shift
    echo "$*" > $P/${name}_cmd.txt
    # (the full bench command under the profiler dies now and then inside hipLaunchKernel -- 3 of 18 runs at the end of round 3, never without
    # the profiler: up to four attempts)
    for attempt in 1 2 3 4; do
        rm -rf $P/$name
        rocprofv3 --kernel-trace --stats --output-format csv -d $P/$name -o run -- "$@" > $P/${name}.log 2>&1 && break
        echo "$name: attempt $attempt died" >> $P/log.txt
    done
}
run_kt kt python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline
run_kt kt1 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --streams 1 --no-batch64 --no-recursion16
run_kt r0 python3 bench.py --shape r0 --width 128 --steps 8 --warmup 2 --no-cpu-baseline --streams 1 --no-batch64
run_kt big21 python3 bench.py --log-n 21 --width 256 --steps 4 --warmup 1 --no-cpu-baseline --streams 1 --no-batch64
run_kt big22 python3 bench.py --log-n 22 --width 128 --steps 4 --warmup 1 --no-cpu-baseline --streams 1 --no-batch64
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -o run -- python3 tools/profile_fused.py > $P/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -o run -- python3 tools/profile_fused.py > $P/pmc_write.log 2>&1
python3 bench.py > $P/bench_default.json 2> $P/bench_default.err
find $P -name "*.csv" | head -40
# keep the transfer small: only the stats tables, counter tables and logs travel back
find $P -name "*kernel_trace.csv" -delete
du -sh $P
