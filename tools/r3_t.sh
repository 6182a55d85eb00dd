#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for t in 16 32 64; do timeout 600 python3 bench.py --steps 4 --warmup 1 --no-batch64 --no-recursion16 --cpu-threads $t 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($t, d['cpu_baseline'])"; done
cat /sys/fs/cgroup/cpu.max
