cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
bash tools/run_profiles_r04.sh > gpurun_out/prof4_run.log 2>&1
tail -30 gpurun_out/prof4_run.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
