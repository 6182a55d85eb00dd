#!/bin/bash
# exploration: per-pass durations inside back-to-back coset LDEs (rocprofv3 kernel trace), default settings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_lde_0 -o run -- python3 tools/lde_loop.py > /dev/null 2>&1
