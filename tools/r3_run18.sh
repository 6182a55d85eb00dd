cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_lockstep.py -m gpu -x -q 2>&1 | tail -25
