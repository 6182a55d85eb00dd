cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_f4
mkdir -p $O
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc1 -o run -- python3 tools/profile_fused.py > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY --output-format csv -d $O/pmc2 -o run -- python3 tools/profile_fused.py > $O/pmc2.log 2>&1
tail -3 $O/pmc1.log $O/pmc2.log
