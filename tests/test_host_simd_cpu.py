"""The host verifiers' batched Poseidon2 (csrc/p2_x16.cpp: 16 queries per AVX-512 register) against the scalar form: the permutation
itself, and the same proofs / corruptions under both settings -- the verdict and the failing check must not depend on it."""
import ctypes as C
import struct

import numpy as np
import pytest

import airs
import machines as M
from zktls_amd import _lib
from zktls_amd._lib import Params
from zktls_amd.device import verify_machine, verify_shard, verify_shard_air

P = 2013265921


def test_batched_permutation_equals_the_scalar_one():
    L = _lib.load()
    a, b = C.c_double(), C.c_double()
    rc = L.zkhip_selftest_host_simd(C.byref(a), C.byref(b))
    assert rc in (0, 1)                                  # 0: no AVX-512 on this CPU
    if rc == 1:
        assert 0 < a.value < b.value * 2                 # not slower than twice the scalar form (it is ~4 x faster)


def test_the_one_register_permutation_is_what_a_transcript_calls():
    """round 6: a transcript is ONE serial sponge chain, so the vector unit is used inside the permutation -- the state in one AVX-512 register (p2_x16.cpp
    p2h_permute behind p2_permute on the host).  zkhip_selftest_host_simd compares it with the scalar form on every state it tries (a mismatch is its -7 return);
    here: it is in use when the CPU has AVX-512, not slower than the scalar form, and switched off together with the batched form"""
    L = _lib.load()
    rc = L.zkhip_selftest_host_simd(None, None)
    assert rc in (0, 1)
    best = lambda form: min(L.zkhip_host_permutation_ns(form) for _ in range(5))      # (the least of five: other processes share this host)
    scalar, fast = best(0), best(1)
    assert scalar > 0 and fast > 0
    if rc == 1:
        assert fast < scalar * 1.1, (fast, scalar)
        prev = L.zkhip_host_simd(0)
        try:
            off = best(1)
        finally:
            L.zkhip_host_simd(prev)
        assert off > fast * 1.1, (off, fast)              # with the vector forms off the transcripts are back on the scalar permutation
    print("host Poseidon2, one state: scalar %.0f ns, one-register %.0f ns" % (scalar, fast))


def verdicts(fn):
    L = _lib.load()
    out = []
    for on in (1, 0):
        prev = L.zkhip_host_simd(on)
        try:
            out.append(fn())
        finally:
            L.zkhip_host_simd(prev)
    return out


def test_same_verdicts_with_and_without_the_batched_form(oracle):
    O = oracle
    cases = []
    t = O.gen_trace(7, 1, 7, 12)
    for shape in ((1, 20, 3), (2, 18, 0, 0, 4, 3, 24), (1, 17, 2, 0, 0, 0, 0, 4)):
        pf = O.prove_shard(t, [1, 2], O.default_params(*shape))
        cases.append(("single", pf, lambda p, shape=shape: verify_shard(p, 7, 12, [1, 2], Params(*shape))))
    fib = airs.fibonacci_program()
    ft, fpub = airs.fibonacci_trace(7, 3, 5)
    pf = O.prove_shard_air(fib, ft, fpub, O.default_params(1, 33, 2))
    cases.append(("air", pf, lambda p: verify_shard_air(fib, p, 7, 4, fpub, Params(1, 33, 2))))
    mt, mp, mtab, mpub = M.range_machine(5, 6)
    lns, ws = [x.shape[0].bit_length() - 1 for x in mt], [x.shape[1] for x in mt]
    pf = O.prove_machine(mt, mp, mtab, mpub, O.default_params(1, 21, 2))
    cases.append(("machine", pf, lambda p: verify_machine(p, lns, ws, mp, mtab, mpub, Params(1, 21, 2))))
    rng = np.random.default_rng(11)
    for name, pf, check in cases:
        assert verdicts(lambda: check(pf)) == [(0, 0), (0, 0)], name
        n_words = pf.size // 4
        for off in [n_words - 3, n_words - 40, n_words // 2, n_words // 3] + rng.integers(40, n_words, 12).tolist():
            bad = bytearray(pf.tobytes())
            v = struct.unpack_from("<I", bad, 4 * int(off))[0]
            struct.pack_into("<I", bad, 4 * int(off), (v + 1) % P)
            arr = np.frombuffer(bytes(bad), dtype=np.uint8)
            a, b = verdicts(lambda: check(arr))
            assert a == b and a[0] == -6, (name, off, a, b)


def test_malformed_input_fuzzer_runs():
    """tests/checks/fuzz_host.py (normally run under the sanitizer build, tools/asan_cpu.sh) for two seconds against the shipped library:
    no crash, and the tool itself stays runnable"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "checks", "fuzz_host.py"), "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout + r.stderr
