cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
for spin in 1 0; do for cfg in "64 1" "16 4"; do ZKHIP_LOCKSTEP_SPIN=$spin timeout 300 python3 tools/lockstep_trace.py $cfg 2>&1 | tail -1; done; done
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
