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


# ---- the chained chip: the initial chaining value is public too; a long message = a chain of shard proofs
def test_chained_program_equals_the_python_restatement(oracle):
    from zktls_amd.device import sha256_air_chained
    prog = S.program(chained=True)
    assert prog.tolist() == sha256_air_chained().tolist()
    assert prog[4] == 32 and oracle.air_validate(prog, S.WIDTH, 32) == 1 and oracle.air_log_quotient_degree(prog) == 1
    assert prog.size == S.program().size + 16                                    # sixteen constants became public-value factors


def test_a_chain_of_two_shards_proven_by_the_oracle(oracle):
    """192 bytes = 4 blocks with padding: two shards of two blocks; shard 1 starts from shard 0's chaining value; the chain's end is the
    digest; each shard's proof is accepted only with ITS (in, out) pair"""
    import hashlib
    from zktls_amd._lib import Params
    from zktls_amd.device import verify_shard_air
    O = oracle
    msg = bytes(range(190))
    blocks = S.pad(msg)
    assert len(blocks) == 256
    t0, out0 = S.trace(blocks[:128])
    iv1 = [out0[2 * k] | (out0[2 * k + 1] << 16) for k in range(8)]
    t1, out1 = S.trace(blocks[128:], chain_in=iv1)
    assert S.digest_bytes(out1) == hashlib.sha256(msg).digest()
    prog = S.program(chained=True)
    iv_limbs = []
    for x in S.IV:
        iv_limbs += [x & 0xffff, x >> 16]
    oprm, prm = O.default_params(1, 5, 3), Params(1, 5, 3)
    p0 = O.prove_shard_air(prog, t0, out0 + iv_limbs, oprm)
    p1 = O.prove_shard_air(prog, t1, out1 + out0, oprm)
    assert verify_shard_air(prog, p0, 7, S.WIDTH, out0 + iv_limbs, prm) == (0, 0)
    assert verify_shard_air(prog, p1, 7, S.WIDTH, out1 + out0, prm) == (0, 0)
    assert verify_shard_air(prog, p1, 7, S.WIDTH, out1 + iv_limbs, prm)[0] == -6          # shard 1 does not start from the standard value
    assert verify_shard_air(prog, p0, 7, S.WIDTH, out1 + iv_limbs, prm)[0] == -6
    # the library's chain verifier on the oracle's proofs
    import ctypes as C
    from zktls_amd import _lib
    L = _lib.load()
    stride = max(p0.size, p1.size)
    buf = np.zeros(2 * stride, dtype=np.uint8)
    buf[:p0.size] = p0
    buf[stride:stride + p1.size] = p1
    lens = (C.c_size_t * 2)(p0.size, p1.size)
    chain = np.array([S.IV, iv1, [out1[2 * k] | (out1[2 * k + 1] << 16) for k in range(8)]], dtype=np.uint32)
    dg = np.frombuffer(hashlib.sha256(msg).digest(), dtype=np.uint8)
    bad, reason = C.c_size_t(0), C.c_int(0)

    def check(chain_, dg_):
        return L.zkhip_verify_sha256_sharded(buf.ctypes.data_as(_lib.u8p), stride, lens, 2, chain_.ctypes.data_as(_lib.u32p), 1, dg_.ctypes.data_as(_lib.u8p),
                                             C.byref(prm), C.byref(bad), C.byref(reason)), bad.value, reason.value
    assert check(chain, dg) == (0, 0, 0)
    other = np.frombuffer(hashlib.sha256(b"x").digest(), dtype=np.uint8)
    assert check(chain, other)[0] == -6
    broken = chain.copy()
    broken[1, 3] ^= 1                                                            # the middle chaining value: both shards fail, the first is named
    assert check(broken, dg)[:2] == (-6, 0)
    shifted = chain.copy()
    shifted[0, 0] ^= 1
    assert check(shifted, dg)[0] == -6                                           # the chain must start from the SHA-256 initial value
    assert L.zkhip_sha256_sharded_count(190, 1) == 2 and L.zkhip_sha256_sharded_count(190, 0) == 4 and L.zkhip_sha256_sharded_count(190, 2) == 1
    assert L.zkhip_sha256_sharded_count(190, 15) == 0
