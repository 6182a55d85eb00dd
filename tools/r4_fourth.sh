cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4e; mkdir -p $O
bash tools/r4_joinprof.sh > $O/joinprof.txt 2>&1; head -30 $O/joinprof.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc; tail -4 $O/pytest.log
