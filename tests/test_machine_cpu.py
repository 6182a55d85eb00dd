"""The machine (proof version 10): chips with their own programs that look each other up through interaction tables -- lookups as
data, with multiplicities and buses.  CPU side: the oracle's prover under the oracle's and the product's verifiers, rejections."""
import struct

import numpy as np
import pytest

import machines as M
from zktls_amd import _lib
from zktls_amd._lib import Params
from zktls_amd.device import verify_machine

P = 2013265921


def shape_of(traces):
    return [t.shape[0].bit_length() - 1 for t in traces], [t.shape[1] for t in traces]


@pytest.mark.parametrize("shape", [(1, 6, 4), (2, 5, 0), (3, 4, 2)])
def test_range_machine_under_both_verifiers(oracle, shape):
    O = oracle
    traces, progs, tables, pub = M.range_machine(5, 6)
    lns, ws = shape_of(traces)
    proof = O.prove_machine(traces, progs, tables, pub, O.default_params(*shape))
    w = np.frombuffer(proof.tobytes(), dtype=np.uint32)
    assert w[1] == 10 and list(w[8:20]) == [7, 4, 1, 1, 6, 4, 1, 3, 5, 4, 1, 1]       # (log_n, width, has-program, interactions) per chip
    assert O.verify_machine(proof, lns, ws, progs, tables, pub, O.default_params(*shape)) == 0
    assert verify_machine(proof, lns, ws, progs, tables, pub, Params(*shape)) == (0, 0)
    # another bus, another multiplicity column, no table: rejected (the header binds the tables' digests)
    for other in ([tables[0], tables[1], O.interaction_table([(O.RECEIVE, 1, 8, [0])])],
                  [tables[0], tables[1], O.interaction_table([(O.RECEIVE, 2, M.BUS_RANGE, [0])])],
                  [tables[0], tables[1], None]):
        assert verify_machine(proof, lns, ws, progs, other, pub, Params(*shape))[0] == -6
        assert O.verify_machine(proof, lns, ws, progs, other, pub, O.default_params(*shape)) != 0
    n_words = proof.size // 4
    rng = np.random.default_rng(n_words)
    for off in sorted(set([10, 21, 50, 70, n_words - 3] + rng.integers(8, n_words, 8).tolist())):
        bad = bytearray(proof.tobytes())
        v = struct.unpack_from("<I", bad, 4 * off)[0]
        struct.pack_into("<I", bad, 4 * off, (v + 1) % P)
        arr = np.frombuffer(bytes(bad), dtype=np.uint8)
        assert verify_machine(arr, lns, ws, progs, tables, pub, Params(*shape))[0] == -6, off
        assert O.verify_machine(arr, lns, ws, progs, tables, pub, O.default_params(*shape)) != 0, off


def test_lookups_that_do_not_balance_are_rejected(oracle):
    O = oracle
    traces, progs, tables, pub = M.range_machine(5, 6)
    lns, ws = shape_of(traces)
    prm, oprm = Params(1, 6, 4), O.default_params(1, 6, 4)
    # a looked-up value outside the table (the USER chip's own constraint still holds)
    bad = [t.copy() for t in traces]
    bad[1][3, 0] = 40
    bad[1][3, 2] = 40 * int(bad[1][3, 1]) % P
    proof = O.prove_machine(bad, progs, tables, pub, oprm)
    assert verify_machine(proof, lns, ws, progs, tables, pub, prm) == (-6, 11) and O.verify_machine(proof, lns, ws, progs, tables, pub, oprm) == 11
    # a wrong multiplicity in the table
    bad = [t.copy() for t in traces]
    bad[2][1, 1] = (int(bad[2][1, 1]) + 1) % P
    proof = O.prove_machine(bad, progs, tables, pub, oprm)
    assert verify_machine(proof, lns, ws, progs, tables, pub, prm)[0] == -6
    # a pair picked twice
    bad = [t.copy() for t in traces]
    row = int(np.flatnonzero(bad[0][:, 2] == 0)[0])
    src = int(np.flatnonzero(bad[0][:, 2] == 1)[0])
    bad[0][row] = bad[0][src]
    proof = O.prove_machine(bad, progs, tables, pub, oprm)
    assert verify_machine(proof, lns, ws, progs, tables, pub, prm)[0] == -6


def test_chips_without_programs_or_tables_join_the_machine(oracle):
    O = oracle
    traces, progs, tables, pub = M.range_machine(5, 6)
    syn = O.gen_trace(5, 1, 5, 8)
    traces, progs, tables = traces + [syn], progs + [None], tables + [None]
    lns, ws = shape_of(traces)
    proof = O.prove_machine(traces, progs, tables, pub, O.default_params(1, 6, 4))
    assert verify_machine(proof, lns, ws, progs, tables, pub, Params(1, 6, 4)) == (0, 0)


def test_malformed_tables_are_refused():
    lib = _lib.load()
    from zktls_amd.device import _program_table
    import oracle_lib as O
    ln, ws = (_lib.C.c_int32 * 1)(6), (_lib.C.c_uint32 * 1)(4)
    kp, pp, pw = _program_table([None])
    good = O.interaction_table([(O.SEND, None, 3, [0, 1])])
    for t in (good, ):
        kt, tp, tw = _program_table([t])
        assert lib.zkhip_machine_proof_size(ln, ws, pp, pw, tp, tw, 1, _lib.C.byref(Params(1, 5, 3)), 0) > 0
    for mutate in (lambda t: t.__setitem__(0, 1), lambda t: t.__setitem__(1, 0), lambda t: t.__setitem__(2, 99), lambda t: t.__setitem__(3, 2),
                   lambda t: t.__setitem__(4, 4), lambda t: t.__setitem__(5, P), lambda t: t.__setitem__(6, 9), lambda t: t.__setitem__(8, 4)):
        t = good.copy()
        mutate(t)
        kt, tp, tw = _program_table([t])
        assert lib.zkhip_machine_proof_size(ln, ws, pp, pw, tp, tw, 1, _lib.C.byref(Params(1, 5, 3)), 0) == 0


@pytest.mark.parametrize("seed", range(10))
def test_random_machines_under_three_verifiers(oracle, seed):
    """tuples of 1..8 values, several buses, multiplicity columns on both sides, odd interaction counts, tables of equal height"""
    import pyverify_chips
    O = oracle
    traces, progs, tables, pub = M.random_machine(seed)
    lns, ws = shape_of(traces)
    shape = (1 + seed % 3, 3, 2)
    proof = O.prove_machine(traces, progs, tables, pub, O.default_params(*shape))
    assert O.verify_machine(proof, lns, ws, progs, tables, pub, O.default_params(*shape)) == 0
    assert verify_machine(proof, lns, ws, progs, tables, pub, Params(*shape)) == (0, 0)
    if seed < 5:
        assert pyverify_chips.verify(proof.tobytes(), lns, ws, pub, *shape, programs=progs, tables=tables) is True
    # one multiplicity off by one somewhere: the sums no longer balance
    c = next(i for i, t in enumerate(tables) if t is not None and int(t[3]) == O.RECEIVE)
    bad = [t.copy() for t in traces]
    mcol = int(tables[c][4])
    bad[c][0, mcol] = (int(bad[c][0, mcol]) + 1) % P
    assert verify_machine(O.prove_machine(bad, progs, tables, pub, O.default_params(*shape)), lns, ws, progs, tables, pub, Params(*shape)) == (-6, 11)


import hashlib
import json
import os

KAT = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_kat.json")))


def golden_machine(g):
    a = g["machine"]
    return M.range_machine(*a[1:]) if a[0] == "range" else M.random_machine(a[1])


@pytest.mark.parametrize("name", sorted(KAT["machine_proofs"]))
def test_golden_machine_proofs(oracle, name):
    """the oracle still produces the committed machine proofs (tests/golden/make_golden.py)"""
    g = KAT["machine_proofs"][name]
    tr, pg, tb, pub = golden_machine(g)
    pf = oracle.prove_machine(tr, pg, tb, pub, oracle.default_params(*g["params"]))
    assert pf.size == g["bytes"] and hashlib.sha256(pf.tobytes()).hexdigest() == g["sha256"]
