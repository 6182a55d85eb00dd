cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_big.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3_t1.log; cat gpurun_out/r3_t1.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err; tail -c 6000 gpurun_out/r3_bench1.json; tail -5 gpurun_out/r3_bench1.err
