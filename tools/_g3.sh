mkdir -p gpurun_out/prof6
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stark.py tests/test_gpu_lockstep.py tests/test_gpu_keyed_machine.py -x -q > gpurun_out/prof6/small_lde_tests.txt 2>&1; tail -15 gpurun_out/prof6/small_lde_tests.txt
timeout 600 python -m pytest tests/test_gpu_recursion_machine.py -x -q -k "sp1" -s > gpurun_out/prof6/sp1_tests.txt 2>&1; tail -25 gpurun_out/prof6/sp1_tests.txt
timeout 300 python tools/lockstep_time.py 64 > gpurun_out/prof6/lockstep_time_small_lde.txt 2>&1; tail -20 gpurun_out/prof6/lockstep_time_small_lde.txt
