"""Chips with their own constraint programs (proof version 9), CPU side: the product's host verifier against the oracle's prover and
verifier on mixed sets (program chips next to synthetic chips), and the rejections."""
import struct

import numpy as np
import pytest

import airs
from zktls_amd import _lib
from zktls_amd._lib import Params
from zktls_amd.device import verify_chips, verify_chips_air

SEED = 0x5A4B544C53
P = 2013265921


def chip_set(oracle, which=0):
    """-> (traces, programs, public values): a counter AIR, a synthetic chip, a Fibonacci AIR -- all over the same 3 public values"""
    O = oracle
    fib = airs.fibonacci_program()
    ft, pub = airs.fibonacci_trace(6, 3, 5)                     # public: a0, b0, b_last
    cnt = airs.counter_program(8).copy()
    cnt[4] = 3                                                  # declared over the shard's three public values (uses 0 and 1)
    ct, _ = airs.counter_trace(8 + which, 8, 3, 5)
    syn = O.gen_trace(SEED, 2, 7, 12)
    return [ct, syn, ft], [cnt, None, fib], pub


@pytest.mark.parametrize("shape", [(1, 6, 4), (2, 5, 0), (3, 4, 3)])
def test_oracle_proofs_under_both_verifiers(oracle, shape):
    O = oracle
    traces, progs, pub = chip_set(O)
    log_ns, widths = [8, 7, 6], [8, 12, 4]
    proof = O.prove_chips_air(traces, progs, pub, O.default_params(*shape))
    w = np.frombuffer(proof.tobytes(), dtype=np.uint32)
    assert w[1] == 9 and list(w[8:17]) == [8, 8, 1, 7, 12, 0, 6, 4, 1]          # (log_n, width, has-program) per chip
    assert (w[17:25] == O.air_digest(progs[0])).all() and (w[25:33] == O.air_digest(progs[2])).all()
    assert O.verify_chips_air(proof, log_ns, widths, progs, pub, O.default_params(*shape)) == 0
    assert verify_chips_air(proof, log_ns, widths, progs, pub, Params(*shape)) == (0, 0)
    lib = _lib.load()
    # other programs, other public values, the plain multi-chip verifier: rejected
    again = [progs[0].copy(), None, airs.fibonacci_program()]                   # equal programs at other addresses: accepted
    assert verify_chips_air(proof, log_ns, widths, again, pub, Params(*shape)) == (0, 0)
    other = [progs[0], None, None]
    assert verify_chips_air(proof, log_ns, widths, other, pub, Params(*shape))[0] == -6
    assert verify_chips_air(proof, log_ns, widths, progs, [pub[0], pub[1], (pub[2] + 1) % P], Params(*shape))[0] == -6
    assert verify_chips(proof, log_ns, widths, pub, Params(*shape))[0] == -6
    n_words = proof.size // 4
    rng = np.random.default_rng(n_words)
    for off in sorted(set([10, 17, 30, 34, n_words - 3] + rng.integers(8, n_words, 8).tolist())):
        bad = bytearray(proof.tobytes())
        v = struct.unpack_from("<I", bad, 4 * off)[0]
        struct.pack_into("<I", bad, 4 * off, (v + 1) % P)
        arr = np.frombuffer(bytes(bad), dtype=np.uint8)
        assert verify_chips_air(arr, log_ns, widths, progs, pub, Params(*shape))[0] == -6, off
        assert O.verify_chips_air(arr, log_ns, widths, progs, pub, O.default_params(*shape)) != 0, off


def test_a_trace_that_breaks_its_program_is_rejected(oracle):
    O = oracle
    traces, progs, pub = chip_set(O)
    bad = traces[0].copy()
    bad[5, 3] = (int(bad[5, 3]) + 1) % P
    proof = O.prove_chips_air([bad] + traces[1:], progs, pub, O.default_params(1, 6, 4))
    assert verify_chips_air(proof, [8, 7, 6], [8, 12, 4], progs, pub, Params(1, 6, 4))[0] == -6


def test_without_programs_the_bytes_are_the_plain_multi_chip_proof(oracle):
    O = oracle
    syn = [O.gen_trace(SEED, 1, 7, 8), O.gen_trace(SEED, 2, 6, 4)]
    a = O.prove_chips_air(syn, [None, None], [1, 2], O.default_params(1, 5, 3))
    assert a.tobytes() == O.prove_chips(syn, [1, 2], O.default_params(1, 5, 3)).tobytes()
    assert verify_chips_air(a, [7, 6], [8, 4], [None, None], [1, 2], Params(1, 5, 3)) == (0, 0)


def test_misuse_is_refused(oracle):
    lib = _lib.load()
    quintic = airs.quintic_program()                              # degree 5: four quotient chunks, which need log_blowup >= 2
    ln, ws = (_lib.C.c_int32 * 1)(6), (_lib.C.c_uint32 * 1)(4)
    from zktls_amd.device import _program_table
    keep, pp, pw = _program_table([quintic])
    assert lib.zkhip_chips_proof_size_air(ln, ws, pp, pw, 1, _lib.C.byref(Params(1, 5, 3)), 1) == 0
    assert lib.zkhip_chips_proof_size_air(ln, ws, pp, pw, 1, _lib.C.byref(Params(2, 5, 3)), 1) > 0
    keep, pp, pw = _program_table([airs.fibonacci_program()])
    assert lib.zkhip_chips_proof_size_air(ln, ws, pp, pw, 1, _lib.C.byref(Params(1, 5, 3)), 2) == 0      # n_public differs from the program's
    assert lib.zkhip_chips_proof_size_air(ln, ws, pp, pw, 1, _lib.C.byref(Params(1, 5, 3)), 3) > 0


@pytest.mark.parametrize("shape", [(2, 5, 3), (3, 4, 0)])
def test_a_chip_of_degree_5_has_four_quotient_chunks(oracle, shape):
    """a table with a degree-5 program next to a degree-3 one and a synthetic one: the header's has-program word carries the program's
    log_quotient_degree (2), that chip opens 16 quotient columns; the oracle's proof under the oracle's, the product's and the Python verifier"""
    import pyverify_chips as V
    from pyverify import Reject
    O = oracle
    qt, qpub = airs.quintic_trace(7, 3)
    lin = O.air_program(4, 1, [(O.SEL_FIRST, [(1, [O.air_var(0)]), (P - 1, [O.air_var(0, public=True)])])])      # degree 2 with its selector
    traces = [qt, O.gen_trace(5, 1, 6, 4), np.full((32, 4), qpub[0], dtype=np.uint32)]
    progs = [airs.quintic_program(), None, lin]
    lns, ws = [7, 6, 5], [4, 4, 4]
    oprm, prm = O.default_params(*shape), Params(*shape)
    proof = O.prove_chips_air(traces, progs, qpub, oprm)
    w = np.frombuffer(proof.tobytes(), dtype=np.uint32)
    assert w[1] == 9 and list(w[8:17]) == [7, 4, 2, 6, 4, 0, 5, 4, 1]
    assert O.verify_chips_air(proof, lns, ws, progs, qpub, oprm) == 0
    assert verify_chips_air(proof, lns, ws, progs, qpub, prm) == (0, 0)
    assert V.verify(proof.tobytes(), lns, ws, qpub, *shape, programs=progs) is True
    n_words = proof.size // 4
    rng = np.random.default_rng(n_words)
    for off in sorted(set([10, 40, 60, 90, n_words - 3] + rng.integers(8, n_words, 12).tolist())):
        bad = bytearray(proof.tobytes())
        v = struct.unpack_from("<I", bad, 4 * off)[0]
        struct.pack_into("<I", bad, 4 * off, (v + 1) % P)
        arr = np.frombuffer(bytes(bad), dtype=np.uint8)
        rc, why = verify_chips_air(arr, lns, ws, progs, qpub, prm)
        assert rc == -6 and why == O.verify_chips_air(arr, lns, ws, progs, qpub, oprm), off
        with pytest.raises(Reject):
            V.verify(bytes(bad), lns, ws, qpub, *shape, programs=progs)
    # the bincode-shaped form carries four chunks for that chip
    L = _lib.load()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    size = L.zkhip_chips_bincode_size(pr.ctypes.data_as(_lib.u8p), pr.size)
    out, got = np.zeros(size, dtype=np.uint8), _lib.C.c_size_t(0)
    pv = np.array(qpub, dtype=np.uint32)
    assert size > 0 and L.zkhip_chips_proof_to_bincode(pr.ctypes.data_as(_lib.u8p), pr.size, pv.ctypes.data_as(_lib.u32p), pv.size, out.ctypes.data_as(_lib.u8p), size, _lib.C.byref(got)) == 0
    back, npub, pub2 = np.zeros(proof.size, dtype=np.uint8), _lib.C.c_size_t(0), np.zeros(4, dtype=np.uint32)
    assert L.zkhip_chips_proof_from_bincode(out.ctypes.data_as(_lib.u8p), size, back.ctypes.data_as(_lib.u8p), back.size, _lib.C.byref(got),
                                            pub2.ctypes.data_as(_lib.u32p), 4, _lib.C.byref(npub)) == 0
    assert back.tobytes() == proof.tobytes()
