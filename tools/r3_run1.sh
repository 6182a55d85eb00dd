cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_f2
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
for G in 0 8192 512; do for R in 0 4; do
echo "== GRID=$G ROT=$R" >> $O/ab.log
FUSED_AB_CHECK=0 ZKHIP_FUSED_GRID=$G ZKHIP_FUSED_ROT=$R python3 tools/fused_ab.py 256 100 --ab 2>&1 | grep -E "which 7|which 6|LDE" >> $O/ab.log
done; done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc1 -o run -- python3 tools/profile_fused.py > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc2 -o run -- python3 tools/profile_fused.py > $O/pmc2.log 2>&1
cat $O/ab.log
ls $O/pmc1 $O/pmc2
