#!/bin/bash
# A/B of the NTT pass kernel variants on 2^20 x 256 (exploration; bench.py is the contract benchmark)
# ZKHIP_NTT_FAST: 4 default tile-per-workgroup kernel, 1 persistent 1024 x 32 kernel; ZKHIP_NTT_CPT: 1 / 2 columns per lane;
# ZKHIP_NTT_MAP: 0 / 1 XCD-aware block map; ZKHIP_NTT_DEBUG: 1 no loads, 2 no stores, 4 no transform
cd /root/repo
run() { echo "== $*"; env "$@" python tools/ntt_pass_time.py 300 2>&1 | tail -1; }
run ZKHIP_NTT_FAST=4
run ZKHIP_NTT_CPT=1
run ZKHIP_NTT_FAST=1
for d in 1 2 3 4; do run ZKHIP_NTT_DEBUG=$d; done
