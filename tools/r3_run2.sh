cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_f3
mkdir -p $O
python3 tools/fused_ab.py 256 200 > $O/ab0.log 2>&1; cat $O/ab0.log
for G in 8192; do for R in 0; do
echo "== GRID=$G ROT=$R" >> $O/ab.log
FUSED_AB_CHECK=0 ZKHIP_FUSED_GRID=$G ZKHIP_FUSED_ROT=$R python3 tools/fused_ab.py 256 100 --ab 2>&1 | grep -E "which 7|which 6|LDE" >> $O/ab.log
done; done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc1 -o run -- python3 tools/profile_fused.py > $O/pmc1.log 2>&1
cat $O/ab.log
