cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/r3_lst
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3_lst -o run -- python3 tools/lockstep_trace.py 11 6 > gpurun_out/r3_lst.log 2>&1
tail -1 gpurun_out/r3_lst.log
python3 tools/window_stats.py $(find gpurun_out/r3_lst -name "*kernel_trace.csv") 75
find gpurun_out/r3_lst -name "*kernel_trace.csv" -delete
