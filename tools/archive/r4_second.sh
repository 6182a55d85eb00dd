# round 4, second GPU call: the shard verifier on the GPU, the fixed logical-device test, hal operators again, the graph-mode repro
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_recursion.py tests/test_gpu_multirank.py tests/test_gpu_hal.py tests/test_gpu_fri_chip.py tests/test_gpu_lockstep.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -8 $O/pytest.log
timeout 300 python3 tools/hal_ops_time.py --out $O/hal_ops.md > $O/hal_ops.log 2>&1; echo "hal rc=$?"; grep -E "mix_poly|batch_eval" $O/hal_ops.md
timeout 600 python3 tools/recursion_time.py > $O/recursion_time.log 2>&1; echo "rec time rc=$?"; tail -12 $O/recursion_time.log
timeout 900 python3 tools/segv/run.py 6 30 nolib_g,nolib_p,lib_a > $O/segv.log 2>&1; cat gpurun_out/segv/summary.txt
