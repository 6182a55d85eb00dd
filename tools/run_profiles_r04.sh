#!/bin/bash
# The rocprofv3 runs behind profiles/r04_* (run on the GPU box through gpurun; outputs under gpurun_out/prof4).
# Kernel trace + stats and the PMC counters are SEPARATE runs (counters only, no trace domains).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/prof4
rm -rf $P; mkdir -p $P
run_kt() {   # name, command...
    local name=$1; shift
    echo "$*" > $P/${name}_cmd.txt
    # (the bench command under the profiler dies now and then inside the profiler's interception of stream operations issued from several
    # host threads -- profiles/r04_segv.md, no-library repro in tools/segv: up to four attempts)
    for attempt in 1 2 3 4; do
        rm -rf $P/$name
        rocprofv3 --kernel-trace --stats --output-format csv -d $P/$name -o run -- "$@" > $P/${name}.log 2>&1 && break
        echo "$name: attempt $attempt died" >> $P/log.txt
    done
}
run_kt kt python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-multichip
run_kt kt1 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --streams 1 --no-batch64 --no-recursion16
run_kt r0 python3 bench.py --shape r0 --width 128 --steps 8 --warmup 2 --no-cpu-baseline --streams 1 --no-batch64
run_kt big21 python3 bench.py --log-n 21 --width 256 --steps 4 --warmup 1 --no-cpu-baseline --streams 1 --no-batch64
run_kt big22 python3 bench.py --log-n 22 --width 128 --steps 4 --warmup 1 --no-cpu-baseline --streams 1 --no-batch64
run_kt join python3 tools/recursion_time.py
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -o run -- python3 tools/profile_fused.py > $P/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -o run -- python3 tools/profile_fused.py > $P/pmc_write.log 2>&1
# the first inverse pass (I1: strided 128-byte row chunks in, one block out) against the in-place forward pass (F2): why I1 sits 5 % lower
for c in TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum \
         TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUBBLE_sum SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $c --output-format csv -d $P/pass_$c -o run -- python3 tools/profile_fused.py > $P/pass_$c.log 2>&1
done
python3 tools/pass_pmc_report.py $P > $P/pass_pmc.md 2>&1; cat $P/pass_pmc.md
python3 bench.py > $P/bench_default.json 2> $P/bench_default.err
find $P -name "*.csv" | head -40
# keep the transfer small: only the stats tables, counter tables and logs travel back
find $P -name "*kernel_trace.csv" -delete
du -sh $P
