"""RISC Zero's group order (code / data / accum / check; reference Cargo.lock:5057 risc0-zkp behind
crates/guest-prover-r0/src/prover.rs:90), proof version 8: oracle proofs under the three verifiers that share no code
(pyverify, the product's host verifier, the oracle's own), the header / size bookkeeping, and rejection of every misuse."""
import struct

import numpy as np
import pytest

import pyverify
from zktls_amd import _lib
from zktls_amd._lib import Params
from zktls_amd.device import verify_shard

SEED = 0x5A4B544C53
# (log_blowup, num_queries, pow_bits, logup_pairs, log_fold, log_final, hash_width, code_width)
SHAPES = [
    (1, 5, 4, 0, 0, 0, 0, 4),          # SP1 FRI shape, two groups
    (2, 4, 0, 2, 4, 2, 24, 8),         # segment shape with lookups: code, data, accum, check
    (1, 5, 4, 1, 0, 0, 0, 12),
    (2, 3, 0, 0, 2, 2, 16, 4),
]


@pytest.mark.parametrize("shape", SHAPES)
def test_three_verifiers_accept_and_reject_together(oracle, shape):
    O = oracle
    log_n, width, pub = 6, 16, [1, 2, 3]
    oprm, prm = O.default_params(*shape), Params(*shape)
    trace = O.gen_trace_logup(SEED, 3, log_n, width, shape[3]) if shape[3] else O.gen_trace(SEED, 3, log_n, width)
    proof = O.prove_shard(trace, pub, oprm)
    lib = _lib.load()
    assert lib.zkhip_proof_size(log_n, width, prm, len(pub)) == proof.size
    w = np.frombuffer(proof.tobytes(), dtype=np.uint32)
    assert w[1] == 8 and w[12] == shape[7]                      # version 8: the 12-word extended header, then code_width
    assert O.verify_shard(proof, log_n, width, pub, oprm) == 0
    assert verify_shard(proof, log_n, width, pub, prm) == (0, 0)
    assert pyverify.verify(proof.tobytes(), log_n, width, pub, *shape) is True
    # the same proof is not a proof for another split, nor for the ungrouped protocol
    for other in (shape[:7] + (shape[7] + 4 if shape[7] + 4 < width else shape[7] - 4,), shape[:7] + (0,)):
        assert verify_shard(proof, log_n, width, pub, Params(*other))[0] == -6
        assert O.verify_shard(proof, log_n, width, pub, O.default_params(*other)) != 0
        with pytest.raises(pyverify.Reject):
            pyverify.verify(proof.tobytes(), log_n, width, pub, *other)
    # single-word corruptions: header, code root, data root, a code path word, the tail
    n_words = proof.size // 4
    H = log_n + shape[0]
    rng = np.random.default_rng(n_words)
    for off in sorted(set([12, 14, 22, n_words - 2, n_words - 8 * H - 3] + rng.integers(13, n_words, 8).tolist())):
        bad = bytearray(proof.tobytes())
        v = struct.unpack_from("<I", bad, 4 * off)[0]
        struct.pack_into("<I", bad, 4 * off, (v + 1) % pyverify.P)
        arr = np.frombuffer(bytes(bad), dtype=np.uint8)
        assert verify_shard(arr, log_n, width, pub, prm)[0] == -6, off
        assert O.verify_shard(arr, log_n, width, pub, oprm) != 0, off
        with pytest.raises(pyverify.Reject):
            pyverify.verify(bytes(bad), log_n, width, pub, *shape)


def test_group_roots_are_the_roots_of_the_column_ranges(oracle):
    """words 13..20 / 21..28 of a version-8 proof = Merkle roots over the code / data columns of the trace LDE"""
    O = oracle
    log_n, width, cw = 6, 16, 4
    shape = (1, 5, 4, 0, 0, 0, 0, cw)
    t = O.gen_trace(SEED, 9, log_n, width)
    proof = O.prove_shard(t, [], O.default_params(*shape))
    w = np.frombuffer(proof.tobytes(), dtype=np.uint32)
    lde = O.coset_lde(t, 1, 31)
    code = O.merkle_tree([np.ascontiguousarray(lde[:, :cw])])[-1]
    data = O.merkle_tree([np.ascontiguousarray(lde[:, cw:])])[-1]
    assert (w[13:21] == code).all() and (w[21:29] == data).all()


def test_misuse_is_refused():
    lib = _lib.load()
    for bad in (3, 16, 20, -4):                                     # not a multiple of 4 / not below the width / negative
        assert lib.zkhip_proof_size(6, 16, Params(1, 5, 4, 0, 0, 0, 0, bad), 0) == 0
    # proof versions 4-7 keep one trace commitment per chip / program
    prm = Params(1, 5, 4, 0, 0, 0, 0, 4)
    ln = (_lib.C.c_int32 * 1)(6)
    ws = (_lib.C.c_uint32 * 1)(16)
    assert lib.zkhip_chips_proof_size(ln, ws, None, None, 1, prm, 0) == 0
    # bincode export covers versions 1-3
    assert lib.zkhip_bincode_size(6, 16, prm) == 0
