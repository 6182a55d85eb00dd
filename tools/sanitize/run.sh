#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (GPU sanitizers are not available on the pool):
#   1. the oracle (plain C) proving and verifying several shapes and a multi-chip shard;
#   2. the PRODUCT's host verifier (libzkhip's C++ host code, built with sanitized host objects) checking the
#      oracle's proofs and single-bit corruptions.  Needs no GPU: zkhip_verify_* never touch the device.
# Everything is built under /tmp.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=/tmp/zkhip_sanitize
mkdir -p $OUT
CLANG=/opt/rocm/lib/llvm/bin/clang
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
ORC="$ROOT/oracle/ntt.c $ROOT/oracle/poseidon2.c $ROOT/oracle/merkle.c $ROOT/oracle/challenger.c $ROOT/oracle/stark.c $ROOT/oracle/chips.c $ROOT/oracle/air.c $ROOT/oracle/hal.c"
$CLANG $SAN -march=x86-64-v3 -Wno-unknown-pragmas -I$ROOT/oracle -o $OUT/oracle_main $ROOT/tools/sanitize/oracle_main.c $ORC -lm
ASAN_OPTIONS=detect_leaks=1 $OUT/oracle_main
for f in ntt.hip ntt_fused.hip hash.hip util.hip stark.hip hal.hip sha256_chip.hip fri_chip.hip context.cpp batch.cpp prover.cpp verifier.cpp jobs.cpp serialize.cpp params.cpp poseidon2_chip.cpp; do
  /opt/rocm/bin/hipcc $SAN -std=c++17 -fPIC --offload-arch=gfx950 -Wno-option-ignored -x hip -c $ROOT/zktls_amd/csrc/$f -o $OUT/$f.o
done
/opt/rocm/bin/hipcc $SAN -std=c++17 -fPIC -mavx512f -mavx512dq -x c++ -c $ROOT/zktls_amd/csrc/p2_x16.cpp -o $OUT/p2_x16.cpp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address,undefined -shared -fPIC -o $OUT/libzkhip_asan.so $OUT/*.o
$CLANG $SAN -march=x86-64-v3 -Wno-unknown-pragmas -I$ROOT/oracle -I$ROOT/include -o $OUT/host_verifier_main $ROOT/tools/sanitize/host_verifier_main.c $ORC \
  -L$OUT -lzkhip_asan -Wl,-rpath,$OUT -lm
ASAN_OPTIONS=detect_leaks=0 $OUT/host_verifier_main
echo "sanitizers: clean"
