cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for Q in "" 8 16; do for F in 4 8 16 32; do
echo "== GPU_MAX_HW_QUEUES=$Q in_flight=$F"
if [ -z "$Q" ]; then python3 tools/batch64_time.py $F 2>&1 | grep -E "keyed SHA-256 machines|proven in" | tail -2
else GPU_MAX_HW_QUEUES=$Q python3 tools/batch64_time.py $F 2>&1 | grep -E "keyed SHA-256 machines|proven in" | tail -2; fi
done; done
