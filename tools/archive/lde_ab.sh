#!/bin/bash
# exploration: per-pass durations inside back-to-back coset LDEs for several tile orders of the block-in / strided-out passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 104 100 102 103 105 106; do
  ZKHIP_NTT_MAP=$m rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_lde_$m -o run -- python3 tools/lde_loop.py > /dev/null 2>&1
done
