mkdir -p gpurun_out/prof6
P=gpurun_out/prof6
timeout 900 python -m pytest tests/test_gpu_recursion_machine.py -x -q 2>&1 | tail -3
python3 tools/tree_breakdown.py 4 > $P/tree_phases_dev.log 2>&1; grep -E "machine verifier|top over|chips prover" $P/tree_phases_dev.log | tail -22 > $P/tree_phases_dev.txt
python3 tools/join_breakdown.py --sha 64 --keyed > $P/keyed64_phases_dev.log 2>&1; tail -22 $P/keyed64_phases_dev.log > $P/keyed64_phases_dev.txt
ZKHIP_REC_HOST=1 python3 tools/join_breakdown.py --sha 64 --keyed > $P/keyed64_phases_host.log 2>&1; tail -22 $P/keyed64_phases_host.log > $P/keyed64_phases_host.txt
echo "== tree dev"; cat $P/tree_phases_dev.txt
echo "== keyed dev"; cat $P/keyed64_phases_dev.txt; echo "== keyed host"; grep -E "machine verifier\]|compress" $P/keyed64_phases_host.txt
