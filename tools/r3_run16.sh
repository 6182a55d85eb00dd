cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 400 python3 tools/lockstep_time.py 64 > gpurun_out/r3_ls64.log 2>&1; echo "rc64=$?"; tail -30 gpurun_out/r3_ls64.log
timeout 900 python3 -m pytest tests/test_gpu_keyed_machine.py tests/test_gpu_parity.py tests/test_gpu_fri_chip.py -m gpu -x -q 2>&1 | tail -5
