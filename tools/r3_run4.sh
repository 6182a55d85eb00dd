cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_f5
mkdir -p $O
python3 tools/fused_ab.py 256 200 --ab > $O/base.log 2>&1; tail -9 $O/base.log
for M in 255 4; do
echo "== persistent passes, ZKHIP_NTT_MAP=$M"
ZKHIP_NTT_FAST=3 ZKHIP_NTT_MAP=$M python3 tools/fused_ab.py 256 200 --ab > $O/pers_$M.log 2>&1; tail -9 $O/pers_$M.log
done
