"""The Poseidon2 chip on the GPU: the on-device trace generator against the Python restatement, proof bytes against the oracle, and the
chip's first use -- openings of the library's OWN Merkle commitment checked in-circuit (what a recursive verifier does per FRI query)."""
import numpy as np
import pytest

import poseidon2_air as A
from zktls_amd._lib import Params, ZkHipError
from zktls_amd.device import p2chip_air, verify_merkle_paths, verify_shard_air

pytestmark = pytest.mark.gpu
SEED = 0x5A4B544C53


@pytest.mark.parametrize("case", [(3, 4, 1), (4, 6, 3), (5, 2, 5), (2, 13, 7)])
def test_trace_equals_the_python_restatement(ctx, case):
    depth, n_paths, seed = case
    leaves, sibs, idx, root = A.tree_paths(depth, n_paths, seed=seed)
    trace, roots = A.merkle_trace(leaves, sibs, idx)
    d, droots, log_n = ctx.p2chip_gen_merkle_trace(leaves, sibs, idx)
    assert log_n == trace.shape[0].bit_length() - 1
    assert (droots == np.array(roots, dtype=np.uint32)).all() and all(r == root for r in roots)
    assert (d.download().reshape(-1, A.WIDTH) == trace).all()
    d.free()


@pytest.mark.parametrize("shape", [(1, 8, 4), (2, 5, 0), (2, 4, 0, 0, 4, 1, 24)])
def test_proof_bytes_equal_the_oracles(ctx, oracle, shape):
    O = oracle
    leaves, sibs, idx, root = A.tree_paths(4, 7, seed=11)
    trace, _ = A.merkle_trace(leaves, sibs, idx)
    prm, oprm = Params(*shape), O.default_params(*shape)
    proof = ctx.prove_merkle_paths(leaves, sibs, idx, root, prm)
    assert proof.tobytes() == O.prove_shard_air(A.program(), trace, root + [7], oprm).tobytes()
    assert verify_merkle_paths(proof, root, 7, prm) == (0, 0)
    assert O.verify_shard_air(A.program(), proof, 5, A.WIDTH, root + [7], oprm) == 0
    # a path that does not end in the root is refused before anything is proven
    wrong = [list(s) for s in sibs]
    wrong[2] = [list(x) for x in wrong[2]]
    wrong[2][1][0] = (wrong[2][1][0] + 1) % O.P
    with pytest.raises(ZkHipError):
        ctx.prove_merkle_paths(leaves, wrong, idx, root, prm)


def test_openings_of_the_librarys_own_commitment_in_circuit(ctx, oracle):
    """commit a 2^12 x 16 matrix with the library (LDE + Poseidon2 Merkle tree), then prove 4096 random openings of that tree through
    the chip: 2^16 rows x 356 columns; the root is the commitment's, the proof is checked by both verifiers"""
    O = oracle
    log_h, n_paths = 13, 4096
    d = ctx.fill_uniform(SEED, log_h - 1, 16)
    lde = ctx.coset_lde(d, log_h - 1, 16)
    tree = ctx.merkle_commit([(lde, 16)], log_h).download().reshape(-1, 8)
    levels, off = [], 0
    for l in range(log_h + 1):
        levels.append(tree[off:off + (1 << (log_h - l))])
        off += 1 << (log_h - l)
    root = levels[-1][0]
    rng = np.random.default_rng(4)
    idx = rng.integers(0, 1 << log_h, n_paths).astype(np.uint32)
    leaves = levels[0][idx]
    sibs = np.stack([levels[l][(idx >> l) ^ 1] for l in range(log_h)], axis=1)
    # the digests of the tree are Montgomery words on the device? no: download() returns canonical words
    prm = Params(1, 30, 8)
    proof = ctx.prove_merkle_paths(leaves, sibs, idx, root, prm)
    assert verify_merkle_paths(proof, root, n_paths, prm) == (0, 0)
    log_n = (n_paths * log_h - 1).bit_length()
    assert O.verify_shard_air(p2chip_air(), proof, log_n, A.WIDTH, root.tolist() + [n_paths], O.default_params(1, 30, 8)) == 0
    assert verify_merkle_paths(proof, root, n_paths - 1, prm)[0] == -6
    # the leaf digest of an opened row is the sponge of that row: the first path's leaf against the oracle's hash of the LDE row
    row = lde.download().reshape(-1, 16)[int(idx[0])]
    assert (O.sponge_hash(row) == leaves[0]).all()


@pytest.mark.parametrize("case", [(3, 3, 8, 1), (2, 5, 24, 2), (4, 2, 64, 3)])
def test_hashed_row_trace_equals_the_python_restatement(ctx, case):
    depth, n_paths, row_width, seed = case
    rng = np.random.default_rng(seed)
    rows_ = rng.integers(0, 2013265921, (n_paths, row_width)).astype(np.uint32)
    sibs = rng.integers(0, 2013265921, (n_paths, depth, 8)).astype(np.uint32)
    idx = rng.integers(0, 1 << depth, n_paths).astype(np.uint32)
    trace, roots = A.merkle_trace(rows_.tolist(), sibs.tolist(), idx.tolist(), hashed_rows=True)
    d, droots, log_n = ctx.p2chip_gen_merkle_trace(rows_, sibs, idx, hashed_rows=True)
    assert (droots == np.array(roots, dtype=np.uint32)).all()
    assert (d.download().reshape(-1, A.WIDTH) == trace).all()
    d.free()


def test_the_trace_openings_of_a_real_proof_checked_in_circuit(ctx, oracle):
    """one step of a recursive verifier on real data: take a shard proof, and prove through the chip what its verifier does for the trace
    commitment of every query -- hash the opened row (sponge) and walk the path to the committed root"""
    from zktls_amd.device import verify_shard
    O = oracle
    log_n, W, Q = 10, 16, 40
    H = log_n + 1
    prm = Params(1, Q, 8)
    t = ctx.gen_trace(SEED, 0, log_n, W)
    proof = ctx.prove_shard(t, log_n, W, [1, 2, 3], prm)
    assert verify_shard(proof, log_n, W, [1, 2, 3], prm) == (0, 0)
    w = np.frombuffer(proof.tobytes(), dtype=np.uint32)
    troot = w[8:16]
    pos = 8 + 16 + 8 * W + 32 + 8 * log_n + 4 + 1                    # header, two roots, openings, FRI commitments, final value, witness
    perq = W + 8 * H + 8 + 8 * H + sum(4 + 8 * (H - 1 - l) for l in range(log_n))
    assert pos + Q * perq == w.size
    lde = ctx.coset_lde(t, log_n, W).download().reshape(-1, W)
    where = {r.tobytes(): i for i, r in enumerate(lde)}
    rows_, sibs, idx = [], [], []
    for q in range(Q):
        base = pos + q * perq
        row = w[base:base + W]
        rows_.append(row)
        sibs.append(w[base + W:base + W + 8 * H].reshape(H, 8))
        idx.append(where[row.tobytes()])                                 # the query's position: where the opened row sits in the LDE
    cproof = ctx.prove_merkle_paths(np.array(rows_), np.array(sibs), np.array(idx, dtype=np.uint32), troot, Params(1, 30, 8), hashed_rows=True)
    assert verify_merkle_paths(cproof, troot, Q, Params(1, 30, 8)) == (0, 0)
    assert O.verify_shard_air(p2chip_air(), cproof, (Q * (W // 8 + H) - 1).bit_length(), A.WIDTH, troot.tolist() + [Q], O.default_params(1, 30, 8)) == 0
    # against another root (the quotient commitment's) the same openings are refused by the prover and by the verifier
    with pytest.raises(ZkHipError):
        ctx.prove_merkle_paths(np.array(rows_), np.array(sibs), np.array(idx, dtype=np.uint32), w[16:24], Params(1, 30, 8), hashed_rows=True)
    assert verify_merkle_paths(cproof, w[16:24], Q, Params(1, 30, 8))[0] == -6


def test_golden_proof_on_gpu(ctx):
    """the committed proof (size, SHA-256 of the bytes) reproduced by the HIP path without the oracle in the loop"""
    import hashlib
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_kat.json")))["p2chip"]
    leaves, sibs, idx, root = A.tree_paths(*g["paths"][:2], seed=g["paths"][2])
    proof = ctx.prove_merkle_paths(leaves, sibs, idx, root, Params(*g["params"]))
    assert proof.size == g["bytes"] and hashlib.sha256(proof.tobytes()).hexdigest() == g["sha256"]
