#!/bin/bash
# AddressSanitizer on the HOST code of libzkhip (verifiers, validators, serialisation, argument checks): builds
# zktls_amd/libzkhip_asan.so (host objects and the host side of fri_chip/sha256_chip/hal.hip built with the address sanitizer, device code as shipped; GPU ASan is not available on this
# pool) and runs the CPU test files that exercise host entries plus tests/checks/fuzz_host.py against it.  usage: tools/asan_cpu.sh [fuzz seconds]
# ThreadSanitizer over the threaded verifiers: make -C zktls_amd/csrc -f asan.mk asan SAN=-fsanitize=thread ASAN_OUT=../libzkhip_tsan.so
# (after rm -rf zktls_amd/csrc/build/asan), then the same python command with LD_PRELOAD=libclang_rt.tsan and that library: the only
# reports come from the oracle's OpenMP runtime (libgomp is not instrumented), none from libzkhip.
set -e
cd "$(dirname "$0")/.."
make -C zktls_amd/csrc -j8 > /dev/null
make -C zktls_amd/csrc -f asan.mk asan
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0
LD_PRELOAD=$RT python -c "
import os, sys
sys.path.insert(0, os.getcwd())
import zktls_amd._lib as l
l.LIB_PATH = os.path.join(os.getcwd(), 'zktls_amd', 'libzkhip_asan.so')
import pytest
sys.exit(pytest.main(['-x', '-q', '-s', '-m', 'not gpu', '-p', 'no:cacheprovider', 'tests/test_pyverify_cpu.py', 'tests/test_air_cpu.py', 'tests/test_serialize_cpu.py', 'tests/test_serialize_chips_cpu.py',
                      'tests/test_groups_cpu.py', 'tests/test_chips_air_cpu.py', 'tests/test_machine_cpu.py', 'tests/test_keyed_machine_cpu.py', 'tests/test_sha256_chip_cpu.py', 'tests/test_abi_cpu.py', 'tests/test_fri_chip_cpu.py', 'tests/test_lockstep_cpu.py', 'tests/test_recursion_cpu.py', 'tests/test_recursion_machine_cpu.py']))
"
ZKHIP_FUZZ_LIB=$PWD/zktls_amd/libzkhip_asan.so LD_PRELOAD=$RT python tests/checks/fuzz_host.py "${1:-30}"
