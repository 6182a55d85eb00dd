#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_fri_chip.py -x -q -k "batch" 2>&1 | tail -4
timeout 600 python3 tools/fri_indices_time.py 5 16 2>&1 | tail -8
rm -rf gpurun_out/fri_prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fri_prof -- python3 tools/fri_indices_time.py 5 16 > gpurun_out/fri_prof.log 2>&1
tail -7 gpurun_out/fri_prof.log
find gpurun_out/fri_prof -name "*kernel_stats.csv" | head -2
find gpurun_out/fri_prof -name "*kernel_trace.csv" -delete
