cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_air.py tests/test_gpu_chips_air.py tests/test_gpu_sha256_chip.py tests/test_gpu_p2chip.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3_t3.log; cat gpurun_out/r3_t3.log
python3 tools/airq_fixed.py > gpurun_out/r3_airq.log 2>&1; cat gpurun_out/r3_airq.log
