mkdir -p gpurun_out/prof6
timeout 900 python -m pytest tests/test_gpu_recursion_machine.py -x -q > gpurun_out/prof6/devwit_tests.txt 2>&1; tail -30 gpurun_out/prof6/devwit_tests.txt
