#!/bin/bash
# The rocprofv3 runs behind profiles/r05_* (run on the GPU box through gpurun; outputs under gpurun_out/prof5).
# Kernel trace + stats and the PMC counters are SEPARATE runs (counters only, no trace domains).  Every attempt is counted in
# $P/attempts.txt (the profiler's interception of multi-threaded stream work dies now and then: profiles/r04_segv.md).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/prof5
rm -rf $P; mkdir -p $P
WHAT=${1:-all}
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
# 1. the multi-chip shard (SP1's real shard structure), one in flight: kernels + the phase table of zkhip_prove_chips (A/B build)
run_kt multichip python3 tools/multichip_breakdown.py
grep -v "^$" $P/multichip.log | tail -60 > $P/multichip_phases.txt
# ... and the same four proofs WITHOUT the profiler (its interception of graph launches inflates the FRI commit phase: 5.2 against 2.2 ms)
python3 tools/multichip_breakdown.py > $P/multichip_plain.log 2>&1; grep -E "chips prover|multichip shard" $P/multichip_plain.log | tail -24 > $P/multichip_phases_plain.txt
# 2. the headline shard alone, one in flight: the clean per-proof table
run_kt kt1 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --streams 1 --no-batch64 --no-recursion16 --no-execution --no-multichip
# 2b. 64 transcript proofs -> ONE proof (air mode of the shard verifier machine): kernels, and the phases (A/B build, no profiler)
run_kt compress64 python3 tools/compress64_trace.py
python3 tools/join_breakdown.py --sha 64 > $P/compress64_phases.log 2>&1; tail -19 $P/compress64_phases.log > $P/compress64_phases.txt
# 2c. the tree: 64 headline shard proofs -> 4 joins of 16 -> one proof (machine mode): the phases of its top (A/B build, no profiler)
run_kt tree python3 tools/tree_breakdown.py 4
python3 tools/tree_breakdown.py 4 > $P/tree_phases.log 2>&1; grep -E "machine verifier|top over|chips prover" $P/tree_phases.log | tail -20 > $P/tree_phases.txt
# 2d. the tree's first level with 1 / 2 / 4 joins in flight (Python threads, then zkhip_prove_shard_verifier_batch): profiles/r05_join_overlap.txt
python3 tools/join_overlap_probe.py 4 16 > $P/join_overlap.txt 2>&1
if [ "$WHAT" = all ]; then
# 3. the contract command (four in flight)
run_kt kt python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-multichip --no-execution
fi
# 4. counters for the streaming passes over the finished LDE (quotient_kernel<4>, rowdot_regs_kernel<4>) and the leaf hash beside them:
# one --pmc pass per counter over six single-shard proofs
for c in FETCH_SIZE WRITE_SIZE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
         TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT; do
  rocprofv3 --pmc $c --output-format csv -d $P/stream_$c -o run -- python3 tools/single_shard_trace.py > $P/stream_$c.log 2>&1 || echo "pmc $c died" >> $P/attempts.txt
done
python3 tools/stream_pmc_report.py $P > $P/stream_pmc.md 2>&1; cat $P/stream_pmc.md
python3 bench.py > $P/bench_default.json 2> $P/bench_default.err
# keep the transfer small: only the stats tables, counter tables and logs travel back
find $P -name "*kernel_trace.csv" -delete
find $P -name "*counter_collection.csv" -size +8M -delete
du -sh $P
cat $P/attempts.txt
