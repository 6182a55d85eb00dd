cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_multirank.py tests/test_server.py tests/test_gpu_configs.py -m gpu -x -q -k "multirank or rank or server or prove_core or config4 or rccl or two_ranks" 2>&1 | tail -25 > gpurun_out/r3_t2.log; cat gpurun_out/r3_t2.log
