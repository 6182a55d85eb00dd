#!/bin/bash
# exploration: signed-Montgomery twiddled butterflies in the NTT pass kernel against the unsigned form (-DNTT_UNSIGNED_BFLY), same box
cd $GRAFT_REPO_ROOT
show() { python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], r['frac'], [v['ms'] for v in r['kernels'].values()], min(r['strided_pass_ms_by_placement']), max(r['strided_pass_ms_by_placement']))"; }
echo "== signed (default)"; show; show
cd zktls_amd/csrc && touch ntt.hip && make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DNTT_UNSIGNED_BFLY" > /dev/null 2>&1; cd ../..
echo "== unsigned"; show; show
