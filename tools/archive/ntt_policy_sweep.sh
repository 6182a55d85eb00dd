#!/bin/bash
# exploration: cache-policy bits of the NTT pass kernel's loads / stores (sc0 = 1, nt = 2, sc1 = 16); builds a sweep library on the box
cd $GRAFT_REPO_ROOT/zktls_amd/csrc
touch ntt.hip
make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DNTT_POLICY_SWEEP" > /dev/null 2>&1 || { echo build failed; exit 1; }
cd ../..
python tools/ntt_policy_ab.py 300
