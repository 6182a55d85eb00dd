mkdir -p gpurun_out/prof6
python3 -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/prof6/smoke.txt 2>&1; tail -2 gpurun_out/prof6/smoke.txt
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/prof6/gpu_suite_final.txt 2>&1; tail -22 gpurun_out/prof6/gpu_suite_final.txt
