# round 4, first GPU call: the whole GPU suite on the changed library, hal operators, the SIGSEGV repro, the leaf hash in situ,
# bench lines (default, r0 shape, one-process)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -5 $O/pytest.log
timeout 300 python3 tools/hal_ops_time.py --out $O/hal_ops.md > $O/hal_ops.log 2>&1; echo "hal rc=$?"
H=gpurun_out/hash_insitu; mkdir -p $H
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$H/trace -o run -- python3 $GRAFT_REPO_ROOT/tools/hash_insitu.py) > $H/trace.log 2>&1
for c in GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$H/$c -o run -- python3 $GRAFT_REPO_ROOT/tools/hash_insitu.py) > $H/$c.log 2>&1
done
python3 tools/hash_insitu_report.py $H > $O/hash_insitu.md 2>&1; tail -12 $O/hash_insitu.md
find $H -name "*.csv" -size +3000k -delete
timeout 900 python3 tools/segv/run.py 6 30 > $O/segv.log 2>&1; cat gpurun_out/segv/summary.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 600 python bench.py --shape r0 --no-cpu-baseline > $O/bench_r0.json 2> $O/bench_r0.err; echo "bench r0 rc=$?"
timeout 600 python bench.py --one-process --no-cpu-baseline --no-batch64 --no-recursion16 > $O/bench_oneproc.json 2> $O/bench_oneproc.err; echo "bench one-process rc=$?"
