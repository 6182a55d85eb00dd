"""The SHA-256 compression chip (csrc/sha256_chip.hip) on the CPU side: the product's constraint program against the test-side
restatement (tests/sha256_air.py, which checks itself row by row in plain integers and against hashlib), FIPS padding, and the
product's host verifier on proofs made by the oracle's generic constraint-program prover."""
import hashlib

import numpy as np
import pytest

import sha256_air as S
from zktls_amd import _lib
from zktls_amd._lib import Params
from zktls_amd.device import sha256_air, sha256_pad, verify_sha256, verify_shard_air


def test_program_words_equal_the_restatement():
    prog = sha256_air()
    ref = S.program()
    assert prog.size == ref.size and (prog == ref).all()
    assert prog[2] == S.WIDTH == 608 and prog[4] == 16
    assert _lib.load().zkhip_air_validate(prog.ctypes.data_as(_lib.u32p), prog.size, 608, 16) == 0


def test_restatement_holds_row_by_row_and_matches_hashlib():
    prog = S.program()
    for msg, total in ((b"", None), (b"abc", None), (bytes(range(119)), None), (bytes(range(150)), None), (b"abc", 4)):
        t, pub = S.trace(S.pad(msg), total)
        assert S.digest_bytes(pub) == hashlib.sha256(msg).digest()
        assert S.check_rows(prog, t, pub) == []
    # a flipped cell breaks some constraint on some row
    t, pub = S.trace(S.pad(b"abc"))
    rng = np.random.default_rng(5)
    for _ in range(12):
        bad = t.copy()
        r, c = int(rng.integers(0, 64)), int(rng.integers(0, 606))
        bad[r, c] = (int(bad[r, c]) + 1) % S.P
        assert S.check_rows(prog, bad, pub), (r, c)


@pytest.mark.parametrize("n", [0, 1, 55, 56, 63, 64, 119, 120, 1000])
def test_padding_is_fips_180_4(n):
    msg = bytes((7 * i + 1) & 0xff for i in range(n))
    assert sha256_pad(msg) == S.pad(msg)


def test_host_verifier_accepts_oracle_proofs_and_rejects_the_wrong_digest(oracle):
    O = oracle
    prog = S.program()
    for msg, shape in ((b"abc", (1, 6, 4)), (bytes(range(150)), (2, 5, 0, 0, 2, 2, 24))):
        t, pub = S.trace(S.pad(msg))
        log_n = t.shape[0].bit_length() - 1
        proof = O.prove_shard_air(prog, t, pub, O.default_params(*shape))
        assert O.verify_shard_air(prog, proof, log_n, S.WIDTH, pub, O.default_params(*shape)) == 0
        digest = hashlib.sha256(msg).digest()
        assert verify_sha256(proof, digest, Params(*shape)) == (0, 0)
        assert verify_shard_air(prog, proof, log_n, S.WIDTH, pub, Params(*shape)) == (0, 0)
        wrong = bytearray(digest)
        wrong[5] ^= 1
        assert verify_sha256(proof, bytes(wrong), Params(*shape))[0] == -6
        # a trace that is not a SHA-256 computation does not verify
        bad = t.copy()
        bad[70 % t.shape[0], S.E + 3] ^= 1
        assert verify_sha256(O.prove_shard_air(prog, bad, pub, O.default_params(*shape)), digest, Params(*shape))[0] == -6
