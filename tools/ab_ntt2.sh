#!/bin/bash
# A/B of the NTT pass kernel variants on 2^20 x 256 (exploration; bench.py is the contract benchmark)
cd /root/repo
run() { echo "== $*"; env "$@" python tools/ntt_pass_time.py 300 2>&1 | tail -1; }
run ZKHIP_NTT_FAST=4
run ZKHIP_NTT_FAST=5
run ZKHIP_NTT_FAST=5 ZKHIP_NTT_MAP=0
run ZKHIP_NTT_FAST=4
run ZKHIP_NTT_FAST=5
