"""The Poseidon2 permutation chip with Merkle-path chaining (csrc/poseidon2_chip.cpp), CPU side: the product's program generator against
the independent Python restatement (tests/poseidon2_air.py, on tests/pyref.py's Poseidon2), the oracle proving the Python-generated trace,
both verifiers, what the chip's constraints catch."""
import ctypes as C

import numpy as np
import pytest

import poseidon2_air as A
from zktls_amd import _lib
from zktls_amd._lib import Params
from zktls_amd.device import p2chip_air, verify_merkle_paths, verify_shard_air

P = 2013265921


def test_program_equals_the_python_restatement(oracle):
    prog = A.program()
    assert prog.tolist() == p2chip_air().tolist()
    assert oracle.air_validate(prog, A.WIDTH, A.N_PUBLIC) == 1 and oracle.air_log_quotient_degree(prog) == 1
    assert A.WIDTH == 360 and prog[3] == 377 and prog.size == 25551


def test_program_follows_the_poseidon2_tables(tmp_path):
    """the round constants are coefficients of the program: another table set, another program (and another digest in every proof)"""
    import json
    import os
    L = _lib.load()
    before = p2chip_air()
    params = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "poseidon2_params.json")))
    f = {"width": 16, "name": "test/p2chip", "external_rc": params["external_rc"], "internal_rc": list(params["internal_rc"]), "internal_diag": params["internal_diag"]}
    f["internal_rc"][3] = (f["internal_rc"][3] + 1) % P
    path = tmp_path / "p2.json"
    path.write_text(json.dumps(f))
    try:
        assert L.zkhip_load_poseidon2_params(str(path).encode()) == 0
        other = p2chip_air()
        assert other.size == before.size and other.tolist() != before.tolist()
    finally:
        L.zkhip_reset_poseidon2_params()
    assert p2chip_air().tolist() == before.tolist()


@pytest.mark.parametrize("shape", [(1, 6, 4), (2, 4, 0)])
def test_paths_proven_by_the_oracle_verify_everywhere(oracle, shape):
    O = oracle
    leaves, sibs, idx, root = A.tree_paths(4, 6, seed=3)
    trace, roots = A.merkle_trace(leaves, sibs, idx)
    assert all(r == root for r in roots) and trace.shape == (32, A.WIDTH)
    prog, pub = A.program(), root + [6]
    oprm, prm = O.default_params(*shape), Params(*shape)
    proof = O.prove_shard_air(prog, trace, pub, oprm)
    assert O.verify_shard_air(prog, proof, 5, A.WIDTH, pub, oprm) == 0
    assert verify_shard_air(prog, proof, 5, A.WIDTH, pub, prm) == (0, 0)
    assert verify_merkle_paths(proof, root, 6, prm) == (0, 0)
    # another count, another root
    assert verify_merkle_paths(proof, root, 5, prm) == (-6, 10)
    assert verify_merkle_paths(proof, [root[0] ^ 1] + root[1:], 6, prm) == (-6, 10)


def test_openings_of_whole_rows_hash_the_leaf_in_circuit(oracle):
    """hashed_rows: a path starts with the sponge rows over the opened row (capacity half chained from row to row), its digest feeds the
    first compression row; the leaf digests equal pyref's sponge_hash"""
    import pyref
    O = oracle
    rng = np.random.default_rng(5)
    depth, n = 3, 3
    rows_ = [[int(v) for v in rng.integers(0, P, 24)] for _ in range(1 << depth)]
    level = [pyref.sponge_hash(r) for r in rows_]
    levels = [level]
    while len(level) > 1:
        level = [pyref.compress(level[2 * i], level[2 * i + 1]) for i in range(len(level) // 2)]
        levels.append(level)
    idx = [5, 0, 6]
    sibs = [[levels[l][(i >> l) ^ 1] for l in range(depth)] for i in idx]
    trace, roots = A.merkle_trace([rows_[i] for i in idx], sibs, idx, hashed_rows=True)
    root = levels[-1][0]
    assert all(r == root for r in roots) and trace.shape == (32, A.WIDTH)
    assert trace[:, A.SS].sum() == 3 and trace[:, A.SPG].sum() == 6 and trace[:, A.CH].sum() == 9 and trace[:, A.END].sum() == 3
    prog, pub = A.program(), root + [n]
    oprm, prm = O.default_params(1, 5, 3), Params(1, 5, 3)
    proof = O.prove_shard_air(prog, trace, pub, oprm)
    assert verify_merkle_paths(proof, root, n, prm) == (0, 0)
    # an opened value changed without re-hashing, a capacity half that does not follow, a first sponge row with a non-zero capacity half
    for row, col in ((1, A.IN + 2), (2, A.IN + 12), (0, A.IN + 9), (7, A.SPG), (6, A.SS)):
        bad = trace.copy()
        bad[row, col] = (int(bad[row, col]) + 1) % P
        assert verify_shard_air(prog, O.prove_shard_air(prog, bad, pub, oprm), 5, A.WIDTH, pub, prm) == (-6, 10), (row, col)


def test_what_the_constraints_catch(oracle):
    """a wrong sibling (the path no longer reaches the root), a broken chain, a forged intermediate, a miscounted END: each makes the AIR
    identity fail at zeta (check 10) although the FRI part of such a proof is fine"""
    O = oracle
    leaves, sibs, idx, root = A.tree_paths(3, 5, seed=8)
    trace, _ = A.merkle_trace(leaves, sibs, idx)
    prog, pub = A.program(), root + [5]
    oprm, prm = O.default_params(1, 5, 3), Params(1, 5, 3)

    def verdict(t):
        return verify_shard_air(prog, O.prove_shard_air(prog, t, pub, oprm), trace.shape[0].bit_length() - 1, A.WIDTH, pub, prm)
    assert verdict(trace) == (0, 0)
    for row, col in ((1, A.IN + 9), (4, A.D + 2), (7, A.X3E(5) + 3), (2, A.SBP(6)), (9, A.SP + 11), (5, A.CNT), (3, A.BIT), (14, A.END), (0, A.CH)):
        bad = trace.copy()
        bad[row, col] = (int(bad[row, col]) + 1) % P
        assert verdict(bad) == (-6, 10), (row, col)
    # a path spliced from two trees: the chain constraint ties a row's digest-carrying half to the previous row's output
    l2, s2, i2, r2 = A.tree_paths(3, 5, seed=9)
    t2, _ = A.merkle_trace(l2, s2, i2)
    spliced = trace.copy()
    spliced[1] = t2[1]
    assert verdict(spliced) == (-6, 10)


def test_entries_check_their_arguments():
    L = _lib.load()
    prm = Params(1, 10, 4)
    assert L.zkhip_merkle_paths_proof_size(0, 4, 0, C.byref(prm)) == 0
    assert L.zkhip_merkle_paths_proof_size(4, 0, 0, C.byref(prm)) == 0
    assert L.zkhip_merkle_paths_proof_size(4, 33, 0, C.byref(prm)) == 0
    assert L.zkhip_merkle_paths_proof_size(1 << 22, 2, 0, C.byref(prm)) == 0
    assert L.zkhip_merkle_paths_proof_size(100, 10, 12, C.byref(prm)) == 0          # an opened row of 12 values: not a multiple of 8
    assert L.zkhip_merkle_paths_proof_size(100, 10, 0, C.byref(prm)) > 0
    assert L.zkhip_merkle_paths_proof_size(100, 10, 256, C.byref(prm)) > L.zkhip_merkle_paths_proof_size(100, 10, 0, C.byref(prm))
    # without a GPU the device entries fail loudly
    z = (C.c_uint32 * 8)()
    got = C.c_size_t(0)
    buf = (C.c_uint8 * 16)()
    assert L.zkhip_prove_merkle_paths(None, z, 0, z, z, 1, 1, z, C.byref(prm), buf, 16, C.byref(got)) != 0
    assert L.zkhip_verify_merkle_paths(buf, 16, z, 1, C.byref(prm), None) != 0


def test_golden_program_digest_and_proof(oracle):
    """the committed fixture (tests/golden/make_golden.py): the program's digest -- the words every proof of the chip carries -- and one
    proof of seven paths, reproduced by the oracle from the Python restatement and accepted by the product's verifier"""
    import hashlib
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_kat.json")))["p2chip"]
    prog = p2chip_air()
    assert prog.size == g["program_words"] and oracle.air_digest(prog).tolist() == g["program_digest"]
    leaves, sibs, idx, root = A.tree_paths(*g["paths"][:2], seed=g["paths"][2])
    assert root == g["root"]
    trace, _ = A.merkle_trace(leaves, sibs, idx)
    pf = oracle.prove_shard_air(A.program(), trace, root + [g["paths"][1]], oracle.default_params(*g["params"]))
    assert pf.size == g["bytes"] and hashlib.sha256(pf.tobytes()).hexdigest() == g["sha256"]
    assert verify_merkle_paths(pf, root, g["paths"][1], Params(*g["params"])) == (0, 0)


def test_the_python_verifier_accepts_the_chips_proofs(oracle):
    """the third, independent verifier (tests/pyverify.py, written from DESIGN.md) on proofs of the two real chips' NEW forms: the Poseidon2
    chip with leaf hashing, and a shard of the chained SHA-256 chip -- both are version-7 proofs of a constraint program"""
    import pyverify
    import sha256_air as S
    O = oracle
    shape = (1, 3, 2)
    oprm = O.default_params(*shape)
    # Poseidon2 chip: two openings of whole rows
    rng = np.random.default_rng(2)
    rows_ = [[int(v) for v in rng.integers(0, P, 16)] for _ in range(4)]
    import pyref
    level = [pyref.sponge_hash(r) for r in rows_]
    l1 = [pyref.compress(level[0], level[1]), pyref.compress(level[2], level[3])]
    root = pyref.compress(l1[0], l1[1])
    idx = [2, 1]
    sibs = [[level[i ^ 1], l1[(i >> 1) ^ 1]] for i in idx]
    trace, roots = A.merkle_trace([rows_[i] for i in idx], sibs, idx, hashed_rows=True)
    assert all(r == root for r in roots)
    proof = O.prove_shard_air(A.program(), trace, root + [2], oprm)
    assert pyverify.verify(proof.tobytes(), 5, A.WIDTH, root + [2], *shape, air=A.program()) is True
    with pytest.raises(pyverify.Reject):
        pyverify.verify(proof.tobytes(), 5, A.WIDTH, root + [3], *shape, air=A.program())
    # chained SHA-256 chip: the second shard of a two-shard message
    blocks = S.pad(bytes(range(100)))
    assert len(blocks) == 128
    t0, out0 = S.trace(blocks[:64], message_len=100, first_block=0)
    iv1 = [out0[2 * k] | (out0[2 * k + 1] << 16) for k in range(8)]
    t1, out1 = S.trace(blocks[64:], chain_in=iv1, message_len=100, first_block=1)
    prog = S.program(chained=True)
    pv1 = S.chained_publics(out1, out0[:16])
    p1 = O.prove_shard_air(prog, t1, pv1, oprm)
    assert pyverify.verify(p1.tobytes(), 6, S.WIDTH, pv1, *shape, air=prog) is True
    with pytest.raises(pyverify.Reject):
        pyverify.verify(p1.tobytes(), 6, S.WIDTH, pv1[:16] + [(pv1[16] + 1) % P] + pv1[17:], *shape, air=prog)
