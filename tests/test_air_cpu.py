"""The AIR as data (constraint programs), host side: the product's validator, digest, synthetic-AIR
program and VERIFIER (zkhip_verify_shard_air, CPU code of libzkhip) against the oracle's prover, plus the independent Python
verifier -- three implementations of the program semantics that share no code.  The device interpreter: tests/test_gpu_air.py."""
import ctypes as C

import numpy as np
import pytest

import airs
import pyverify
from zktls_amd import _lib
from zktls_amd._lib import Params, u32p
from zktls_amd.device import air_synthetic, verify_shard_air

P = 2013265921
SEED = 0x5A4B544C53


def test_synthetic_program_digest_and_validation_agree_with_the_oracle(oracle):
    L = _lib.load()
    for width, npub in ((4, 0), (8, 3), (64, 9)):
        prog = air_synthetic(width, npub)
        assert (prog == oracle.air_synthetic(width, npub)).all()
        assert oracle.air_validate(prog, width, npub) == 1
        assert L.zkhip_air_validate(prog.ctypes.data_as(u32p), prog.size, width, npub) == 0
        d = np.zeros(8, dtype=np.uint32)
        assert L.zkhip_air_digest(prog.ctypes.data_as(u32p), prog.size, d.ctypes.data_as(u32p)) == 0
        assert (d == oracle.air_digest(prog)).all() and d.tolist() == pyverify.air_digest(prog)


def test_malformed_programs_are_refused_everywhere(oracle):
    L = _lib.load()
    good = airs.fibonacci_program()

    def both(prog, width=4, npub=3):
        prog = np.ascontiguousarray(prog, dtype=np.uint32)
        a = L.zkhip_air_validate(prog.ctypes.data_as(u32p), prog.size, width, npub) == 0
        b = oracle.air_validate(prog, width, npub) == 1
        assert a == b
        return a
    assert both(good)
    for mutate in (lambda p: p.__setitem__(0, 1), lambda p: p.__setitem__(2, 8), lambda p: p.__setitem__(5, p[5] + 1),
                   lambda p: p.__setitem__(6, 4),                       # selector out of range
                   lambda p: p.__setitem__(8, P),                       # non-canonical coefficient
                   lambda p: p.__setitem__(9, 6),                       # degree 6
                   lambda p: p.__setitem__(10, (1 << 30) | 4),          # column out of range
                   lambda p: p.__setitem__(10, (2 << 30) | 3),          # public value out of range
                   lambda p: p.__setitem__(10, (3 << 30))):             # unknown variable kind
        bad = good.copy()
        mutate(bad)
        assert not both(bad)
    assert not both(good[:-1]) and not both(np.concatenate([good, [0]])) and not both(good, width=8) and not both(good, npub=2)


@pytest.mark.parametrize("shape", [(1, 6, 4, 0, 0, 0, 0), (2, 5, 0, 0, 4, 2, 24), (1, 4, 3, 0, 2, 0, 16)])
def test_three_verifiers_agree_on_program_proofs(oracle, shape):
    prm, oprm = Params(*shape), oracle.default_params(*shape)
    cases = []
    t, pub = airs.fibonacci_trace(6, 3, 5)
    cases.append((airs.fibonacci_program(), t, pub, 6, 4))
    t, pub = airs.counter_trace(6, 8, 77, 5)
    cases.append((airs.counter_program(8), t, pub, 6, 8))
    cases.append((air_synthetic(8, 2), oracle.gen_trace(SEED, 1, 6, 8), [4, 5], 6, 8))
    for prog, trace, pub, log_n, width in cases:
        proof = oracle.prove_shard_air(prog, trace, pub, oprm)
        assert oracle.verify_shard_air(prog, proof, log_n, width, pub, oprm) == 0
        assert verify_shard_air(prog, proof, log_n, width, pub, prm) == (0, 0)
        assert pyverify.verify(proof.tobytes(), log_n, width, pub, *shape, air=prog) is True
        # bound to the program, the public values and the proof words
        other = air_synthetic(width, len(pub)) if prog is not cases[2][0] else airs.counter_program(8)
        assert verify_shard_air(other, proof, log_n, width, pub, prm)[0] == -6
        with pytest.raises(pyverify.Reject):
            pyverify.verify(proof.tobytes(), log_n, width, pub, *shape, air=other)
        wrong = list(pub)
        wrong[-1] = (wrong[-1] + 1) % P
        assert verify_shard_air(prog, proof, log_n, width, wrong, prm)[0] == -6
        assert oracle.verify_shard_air(prog, proof, log_n, width, wrong, oprm) != 0
        bad = proof.copy().view(np.uint32)
        bad[30] = (int(bad[30]) + 1) % P
        assert verify_shard_air(prog, bad.view(np.uint8), log_n, width, pub, prm)[0] == -6
        with pytest.raises(pyverify.Reject):
            pyverify.verify(bad.tobytes(), log_n, width, pub, *shape, air=prog)


@pytest.mark.parametrize("shape", [(2, 5, 4, 0, 0, 0, 0), (3, 4, 0, 0, 2, 0, 24)])
def test_degree_five_programs_use_four_quotient_chunks(oracle, shape):
    """log_quotient_degree 2 (SURVEY.md 8a row a9 lists {1, 2}): the quotient domain is 4N points, four chunks of 4 base columns"""
    prog = airs.quintic_program()
    assert oracle.air_log_quotient_degree(prog) == 2 and pyverify.air_log_quotient_degree(prog) == 2
    t, pub = airs.quintic_trace(6, 9)
    prm, oprm = Params(*shape), oracle.default_params(*shape)
    proof = oracle.prove_shard_air(prog, t, pub, oprm)
    assert oracle.verify_shard_air(prog, proof, 6, 4, pub, oprm) == 0
    assert verify_shard_air(prog, proof, 6, 4, pub, prm) == (0, 0)
    assert pyverify.verify(proof.tobytes(), 6, 4, pub, *shape, air=prog) is True
    assert verify_shard_air(prog, proof, 6, 4, [8], prm)[0] == -6
    bad = proof.copy().view(np.uint32)
    bad[60] = (int(bad[60]) + 1) % P                     # one of the 16 quotient openings
    assert verify_shard_air(prog, bad.view(np.uint8), 6, 4, pub, prm)[0] == -6
    with pytest.raises(pyverify.Reject):
        pyverify.verify(bad.tobytes(), 6, 4, pub, *shape, air=prog)
    # blowup 2 cannot hold a 4N-point quotient domain
    assert verify_shard_air(prog, proof, 6, 4, pub, Params(1, shape[1], shape[2]))[0] == -6
    with pytest.raises(RuntimeError):
        oracle.prove_shard_air(prog, t, pub, oracle.default_params(1, 5, 4))


def test_a_violating_trace_never_yields_an_accepted_proof(oracle):
    prog = airs.fibonacci_program()
    t, pub = airs.fibonacci_trace(6, 3, 5)
    prm, oprm = Params(1, 6, 4), oracle.default_params(1, 6, 4)
    for (r, c) in ((10, 2), (10, 0), (63, 1), (0, 0), (5, 3)):
        t2 = t.copy()
        t2[r, c] = (int(t2[r, c]) + 1) % P
        try:
            proof = oracle.prove_shard_air(prog, t2, pub, oprm)
        except RuntimeError:
            continue
        assert verify_shard_air(prog, proof, 6, 4, pub, prm)[0] == -6
        with pytest.raises(pyverify.Reject):
            pyverify.verify(proof.tobytes(), 6, 4, pub, 1, 6, 4, air=prog)
