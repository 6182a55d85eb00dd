cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 0 64 128; do
  ZKHIP_NTT_DEBUG=$d rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_lde_$d -o run -- python3 tools/lde_loop.py > /dev/null 2>&1
done
