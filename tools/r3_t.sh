#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python3 -m pytest tests/test_gpu_lockstep.py tests/test_gpu_fri_chip.py tests/test_gpu_keyed_machine.py tests/test_gpu_examples.py -x -q 2>&1 | tail -3
timeout 300 python3 tools/fri_indices_time.py 3 16 2>&1 | tail -2
./examples/compress_shards 64 14 64 | tail -1
