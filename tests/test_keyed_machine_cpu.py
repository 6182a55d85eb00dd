"""The keyed machine (proof version 11): chips with PREPROCESSED columns committed once by setup (the reference's `client.setup`,
crates/guest-prover-sp1/src/sp1.rs:113; sp1-stark StarkMachine::setup).  CPU side: the oracle's setup and prover under the oracle's and
the product's verifiers, the rejections either gives, the argument checks of the C entries."""
import ctypes as C
import hashlib
import json
import os
import struct

import numpy as np
import pytest

import machines as M
from zktls_amd import _lib
from zktls_amd._lib import Params
from zktls_amd.device import verify_machine_keyed

P = 2013265921
KAT = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_kat.json")))


def golden_keyed_machine(g):
    a = g["machine"]
    return M.byte_machine(*a[1:]) if a[0] == "byte" else M.random_keyed_machine(a[1])


@pytest.mark.parametrize("name", sorted(KAT["keyed_machine_proofs"]))
def test_golden_keyed_machine_proofs(oracle, name):
    """the oracle still produces the committed keys and proofs (tests/golden/make_golden.py); the product's verifier accepts them"""
    g = KAT["keyed_machine_proofs"][name]
    tr, pre, pg, tb, pub = golden_keyed_machine(g)
    lns, ws, pws = shape_of(tr, pre)
    oprm = oracle.default_params(*g["params"])
    assert oracle.machine_setup(pre, lns, oprm).tolist() == g["root"]
    pf = oracle.prove_machine_keyed(tr, pre, pg, tb, pub, oprm)
    assert pf.size == g["bytes"] and hashlib.sha256(pf.tobytes()).hexdigest() == g["sha256"]
    assert verify_machine_keyed(pf, lns, ws, pws, g["root"], pg, tb, pub, Params(*g["params"])) == (0, 0)


def shape_of(traces, pre):
    return ([t.shape[0].bit_length() - 1 for t in traces], [t.shape[1] for t in traces], [0 if p is None else p.shape[1] for p in pre])


@pytest.mark.parametrize("shape", [(1, 6, 4), (2, 5, 0), (3, 4, 2)])
def test_byte_machine_under_both_verifiers(oracle, shape):
    O = oracle
    traces, pre, progs, tables, pub = M.byte_machine(6, 3)
    lns, ws, pws = shape_of(traces, pre)
    oprm, prm = O.default_params(*shape), Params(*shape)
    root = O.machine_setup(pre, lns, oprm)
    proof = O.prove_machine_keyed(traces, pre, progs, tables, pub, oprm)
    w = np.frombuffer(proof.tobytes(), dtype=np.uint32)
    # header entries (log_n, width, has-program, interactions, preprocessed width); the key's root after the four digests
    assert w[1] == 11 and list(w[8:18]) == [6, 4, 1, 1, pws[0], 6, 4, 1, 1, pws[1]] and sorted(pws) == [0, 4]
    assert list(w[18 + 32:18 + 40]) == list(root)
    assert O.verify_machine_keyed(proof, lns, ws, pws, root, progs, tables, pub, oprm) == 0
    assert verify_machine_keyed(proof, lns, ws, pws, root, progs, tables, pub, prm) == (0, 0)
    # under another key, with other preprocessed widths, as a plain machine: refused
    other = root.copy()
    other[3] = (int(other[3]) + 1) % P
    assert verify_machine_keyed(proof, lns, ws, pws, other, progs, tables, pub, prm) == (-6, 3)
    assert O.verify_machine_keyed(proof, lns, ws, pws, other, progs, tables, pub, oprm) == 3
    assert verify_machine_keyed(proof, lns, ws, pws[::-1], root, progs, tables, pub, prm)[0] == -6
    # every word of the proof matters, and both verifiers name the same check
    n_words = proof.size // 4
    rng = np.random.default_rng(n_words)
    seen = set()
    for off in sorted(set([9, 12, 55, 70, 90, n_words - 3] + rng.integers(8, n_words, 40).tolist())):
        bad = bytearray(proof.tobytes())
        v = struct.unpack_from("<I", bad, 4 * off)[0]
        struct.pack_into("<I", bad, 4 * off, (v + 1) % P)
        arr = np.frombuffer(bytes(bad), dtype=np.uint8)
        rc, why = verify_machine_keyed(arr, lns, ws, pws, root, progs, tables, pub, prm)
        assert rc == -6 and why == O.verify_machine_keyed(arr, lns, ws, pws, root, progs, tables, pub, oprm), off
        seen.add(why)
    assert 33 in seen or shape[1] < 5          # a preprocessed row that does not open the key's root


def test_the_key_fixes_the_table(oracle):
    """the table's contents are bound by the key alone: a prover with another table produces a proof the honest key refuses,
    and a lookup of a tuple the table does not hold does not balance"""
    O = oracle
    traces, pre, progs, tables, pub = M.byte_machine(6, 3)
    lns, ws, pws = shape_of(traces, pre)
    oprm, prm = O.default_params(1, 6, 4), Params(1, 6, 4)
    root = O.machine_setup(pre, lns, oprm)
    forged = [None if p is None else p.copy() for p in pre]
    k = [i for i, p in enumerate(pre) if p is not None][0]
    forged[k][5, 2] ^= 1                                          # "a XOR b" wrong in one row of the table
    proof = O.prove_machine_keyed(traces, forged, progs, tables, pub, oprm)
    assert verify_machine_keyed(proof, lns, ws, pws, root, progs, tables, pub, prm) == (-6, 3)
    assert verify_machine_keyed(proof, lns, ws, pws, O.machine_setup(forged, lns, oprm), progs, tables, pub, prm) in ((0, 0), (-6, 11))
    # a user row that claims a wrong XOR: its tuple is not in the table, the sums do not cancel
    bad = [t.copy() for t in traces]
    u = 1 - k
    bad[u][3, 2] ^= 1
    proof = O.prove_machine_keyed(bad, pre, progs, tables, pub, oprm)
    assert O.verify_machine_keyed(proof, lns, ws, pws, root, progs, tables, pub, oprm) == 11
    assert verify_machine_keyed(proof, lns, ws, pws, root, progs, tables, pub, prm) == (-6, 11)
    # a wrong multiplicity in the table's main column
    bad = [t.copy() for t in traces]
    bad[k][9, 0] = (int(bad[k][9, 0]) + 1) % P
    proof = O.prove_machine_keyed(bad, pre, progs, tables, pub, oprm)
    assert verify_machine_keyed(proof, lns, ws, pws, root, progs, tables, pub, prm) == (-6, 11)


def test_keyed_entries_check_their_arguments(oracle):
    O = oracle
    L = _lib.load()
    traces, pre, progs, tables, pub = M.byte_machine(6, 3)
    lns, ws, pws = shape_of(traces, pre)
    n = len(lns)
    ln = (C.c_int32 * n)(*lns); w_ = (C.c_uint32 * n)(*ws)
    from zktls_amd.device import _program_table
    kp, pp, pw = _program_table(progs)
    kt, tp, tw = _program_table(tables)
    prm = Params(1, 6, 4)
    good = (C.c_uint32 * n)(*pws)
    assert L.zkhip_machine_proof_size_keyed(ln, w_, good, pp, pw, tp, tw, n, C.byref(prm), 1) > 0
    assert L.zkhip_machine_proof_size_keyed(ln, w_, good, pp, pw, tp, tw, n, C.byref(prm), 1) == \
        O.prove_machine_keyed(traces, pre, progs, tables, pub, O.default_params(1, 6, 4)).size
    # no preprocessed columns at all, a width that is not a multiple of 4, programs of the wrong (not combined) width
    for bad in ([0] * n, [2 if p else 0 for p in pws], pws[::-1]):
        assert L.zkhip_machine_proof_size_keyed(ln, w_, (C.c_uint32 * n)(*bad), pp, pw, tp, tw, n, C.byref(prm), 1) == 0
    # a chip with preprocessed columns but no program of its own
    k = [i for i, p in enumerate(pws) if p][0]
    kp2, pp2, pw2 = _program_table([None if i == k else p for i, p in enumerate(progs)])
    assert L.zkhip_machine_proof_size_keyed(ln, w_, good, pp2, pw2, tp, tw, n, C.byref(prm), 1) == 0
    # without a GPU the device entries fail loudly
    key = C.c_void_p()
    root = (C.c_uint32 * 8)()
    assert L.zkhip_machine_setup(None, None, n, C.byref(prm), C.byref(key), root) != 0
    L.zkhip_machine_key_destroy(None)


def test_transcript_batch_without_a_gpu_fails_loudly():
    """zkhip_prove_transcripts on a box without a device: every job marked NO_DEVICE, nothing proven on the CPU"""
    from zktls_amd import _lib as lib_mod
    L = lib_mod.load()
    if lib_mod.device_count() > 0:
        pytest.skip("a GPU is present")
    jobs = (lib_mod.TranscriptJob * 2)()
    msg = np.frombuffer(b"transcript", dtype=np.uint8)
    buf = np.zeros(16, dtype=np.uint8)
    for j in jobs:
        j.message = msg.ctypes.data_as(lib_mod.u8p); j.message_len = msg.size; j.proof = buf.ctypes.data_as(lib_mod.u8p); j.proof_cap = 16
    vk = (C.c_uint32 * 8)()
    prm = Params(1, 10, 4)
    assert L.zkhip_prove_transcripts(None, 0, jobs, 2, C.byref(prm), 4, 1, vk) == -2        # ZKHIP_ERR_NO_DEVICE
    assert [j.status for j in jobs] == [-2, -2] and all(j.proof_len == 0 for j in jobs)
    assert L.zkhip_prove_transcripts(None, 0, jobs, 0, C.byref(prm), 4, 0, vk) == 0           # an empty batch is fine anywhere
    assert L.zkhip_prove_transcripts(None, 3, jobs, 2, C.byref(prm), 4, 0, vk) == -1          # NULL device list with a count
    assert L.zkhip_prove_transcripts((C.c_int * 2)(0, 0), 2, jobs, 2, C.byref(prm), 4, 0, vk) == -1   # a device listed twice
