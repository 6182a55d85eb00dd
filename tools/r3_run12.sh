cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python3 tools/lockstep_time.py 16 > gpurun_out/r3_ls16.log 2>&1; echo "rc16=$?"; tail -25 gpurun_out/r3_ls16.log
timeout 600 python3 tools/lockstep_time.py 64 > gpurun_out/r3_ls64.log 2>&1; echo "rc64=$?"; tail -25 gpurun_out/r3_ls64.log
timeout 900 python3 -m pytest tests/test_gpu_keyed_machine.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
