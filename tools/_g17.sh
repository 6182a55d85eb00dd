mkdir -p gpurun_out/prof6
P=gpurun_out/prof6
timeout 1200 python -m pytest tests/test_gpu_recursion.py tests/test_gpu_recursion_machine.py tests/test_gpu_fri_chip.py tests/test_gpu_examples.py -x -q 2>&1 | tail -4
python3 tools/join_breakdown.py --sha 64 > $P/compress64_phases.log 2>&1; tail -21 $P/compress64_phases.log > $P/compress64_phases.txt
python3 tools/join_breakdown.py --keyed 64 > $P/keyed64_phases.log 2>&1; tail -23 $P/keyed64_phases.log > $P/keyed64_phases.txt
python3 tools/tree_breakdown.py 4 > $P/tree_phases.log 2>&1; grep -E "machine verifier|top over|chips prover" $P/tree_phases.log | tail -23 > $P/tree_phases.txt
echo == c64; head -12 $P/compress64_phases.txt; echo == k64; grep -E "machine verifier\]|compress" $P/keyed64_phases.txt; echo == tree; grep -E "machine verifier\]|top over" $P/tree_phases.txt; grep "shard verifier\]" $P/tree_phases.log | tail -8
