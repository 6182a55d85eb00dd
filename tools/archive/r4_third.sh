# round 4, third GPU call: the join on the GPU, the SIGSEGV repro with the graph switch, timings, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_recursion.py tests/test_gpu_fri_chip.py -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -8 $O/pytest.log
timeout 600 python3 tools/recursion_time.py > $O/recursion_time.log 2>&1; echo "rec time rc=$?"; tail -8 $O/recursion_time.log
timeout 900 python3 tools/segv/run.py 6 30 lib_e,lib_d,nolib_g > $O/segv.log 2>&1; cat gpurun_out/segv/summary.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
