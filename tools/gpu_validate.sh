# the full validation of a tree on a GPU box (through gpurun): the GPU test suite, the default bench line, smoke(), the shard verifier at the headline size
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/validate; mkdir -p $O
timeout 1500 python3 -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cat $O/bench.json | head -c 6000
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 300 python3 tools/recursion_time.py > $O/rec.log 2>&1; tail -12 $O/rec.log
