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
    assert prog[2] == S.WIDTH == 640 and prog[4] == S.N_PUBLIC == 91
    assert _lib.load().zkhip_air_validate(prog.ctypes.data_as(_lib.u32p), prog.size, 640, 91) == 0


@pytest.mark.parametrize("n", [0, 1, 3, 54, 55, 56, 57, 63, 64, 65, 119, 120, 127, 128, 1000, 70000])
def test_padding_publics_equal_the_restatement(n):
    """the 75 public values a verifier derives from the message length: the library's function against the restatement's, for whole messages and
    for every two-block slice of them (the chained shards)"""
    from zktls_amd.device import sha256_padding_publics
    assert sha256_padding_publics(n).tolist() == S.padding_publics(n)[0]
    k = (n + 8) // 64 + 1
    for first in range(0, min(k, 12), 2):
        act = min(2, k - first)
        assert sha256_padding_publics(n, first, act).tolist() == S.padding_publics(n, first, act)[0], (n, first)


def test_restatement_holds_row_by_row_and_matches_hashlib():
    prog = S.program()
    for msg, total in ((b"", None), (b"abc", None), (bytes(range(119)), None), (bytes(range(150)), None), (b"abc", 4), (bytes(55), None), (bytes(56), None), (bytes(range(63)), 4), (bytes(64), None), (bytes(range(121)), 8)):
        t, pub = S.trace(S.pad(msg), total)
        assert S.digest_bytes(pub) == hashlib.sha256(msg).digest()
        assert S.check_rows(prog, t, pub) == []
    # a flipped cell breaks some constraint on some row
    t, pub = S.trace(S.pad(b"abc"))
    rng = np.random.default_rng(5)
    for _ in range(12):
        bad = t.copy()
        r, c = int(rng.integers(0, 64)), int(rng.integers(0, S.Z2 + 1))            # (the columns behind Z2 pad the row to whole 32-column tiles: no constraint reads them)
        bad[r, c] = (int(bad[r, c]) + 1) % S.P
        assert S.check_rows(prog, bad, pub), (r, c)


def test_the_statement_is_sha256_of_a_message_of_public_length():
    """what the padding constraints refuse (row by row, plain integers).  Until round 4 each of these traces satisfied the
    program: ACT dropped one block early, another length field, a missing 0x80, a message that runs on where zeros must be, one more block"""
    prog = S.program()
    msg = bytes(range(150))                                  # 3 blocks: the 0x80 byte and the length share the last one
    t, pub = S.trace(S.pad(msg), 4)
    assert S.check_rows(prog, t, pub) == []

    def forged(blocks, message_len, total=4):
        """an HONEST trace generator run on other blocks / another claimed length: every structural column is consistent with ITS claim"""
        return S.trace(blocks, total, message_len=message_len)

    # (a) ACT drops one block early: the chain over the first two blocks only, claimed against the 150-byte statement's public values
    t2, pub2 = forged(S.pad(msg)[:128], 150 - 64)            # (an honest trace of a two-block chain ...)
    assert S.check_rows(prog, t2, pub2[:16] + pub[16:])      # ... is refused under the three-block statement (CNT, the flags, the padding)
    # the same with the flags forged to the three-block pattern but ACT dropped early
    early = t.copy()
    early[128:192, S.ACT] = 0
    assert S.check_rows(prog, early, pub)
    # (b) a wrong length field in the last block / a statement with another length
    other_len, _ = S.padding_publics(151)
    assert S.check_rows(prog, t, pub[:16] + other_len)
    # (c) a missing 0x80: blocks whose boundary byte is zero, everything else as the honest generator writes it
    blocks = bytearray(S.pad(msg))
    blocks[150] = 0
    t3, pub3 = forged(bytes(blocks), 150)
    assert S.check_rows(prog, t3, pub3)
    # (d) message bytes where zeros must be (between the 0x80 byte and the length field)
    blocks = bytearray(S.pad(msg))
    blocks[160] = 7
    t4, pub4 = forged(bytes(blocks), 150)
    assert S.check_rows(prog, t4, pub4)
    # (e) one block MORE than the length allows (a fourth active block)
    t5, pub5 = S.trace(bytes(S.pad(msg)) + bytes(64), 4, message_len=150)
    assert S.check_rows(prog, t5, pub5[:16] + pub[16:])
    # (f) the 0x80 byte in the block BEFORE the length block (L mod 64 >= 56): honest holds, a message byte in the length block does not
    msg2 = bytes(range(60))
    t6, pub6 = S.trace(S.pad(msg2))
    assert t6.shape[0] == 128 and S.check_rows(prog, t6, pub6) == []
    blocks = bytearray(S.pad(msg2))
    blocks[64 + 5] = 1
    t7, pub7 = forged(bytes(blocks), 60, 2)
    assert S.check_rows(prog, t7, pub7)


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
        assert verify_sha256(proof, digest, Params(*shape), len(msg)) == (0, 0)
        assert verify_sha256(proof, digest, Params(*shape), len(msg) + 1)[0] == -6           # the same digest claimed for another length
        assert verify_shard_air(prog, proof, log_n, S.WIDTH, pub, Params(*shape)) == (0, 0)
        wrong = bytearray(digest)
        wrong[5] ^= 1
        assert verify_sha256(proof, bytes(wrong), Params(*shape), len(msg))[0] == -6
        # a trace that is not a SHA-256 computation does not verify
        bad = t.copy()
        bad[70 % t.shape[0], S.E + 3] ^= 1
        assert verify_sha256(O.prove_shard_air(prog, bad, pub, O.default_params(*shape)), digest, Params(*shape), len(msg))[0] == -6


# ---- the chained chip: the initial chaining value is public too; a long message = a chain of shard proofs
def test_chained_program_equals_the_python_restatement(oracle):
    from zktls_amd.device import sha256_air_chained
    prog = S.program(chained=True)
    assert prog.tolist() == sha256_air_chained().tolist()
    assert prog[4] == S.N_PUBLIC_CHAINED == 107 and oracle.air_validate(prog, S.WIDTH, 107) == 1 and oracle.air_log_quotient_degree(prog) == 1
    assert prog.size == S.program().size + 16 and oracle.air_log_quotient_degree(S.program()) == 1      # sixteen constants became public-value factors; the padding keeps to degree 3


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
    t0, out0 = S.trace(blocks[:128], message_len=190, first_block=0)
    iv1 = [out0[2 * k] | (out0[2 * k + 1] << 16) for k in range(8)]
    t1, out1 = S.trace(blocks[128:], chain_in=iv1, message_len=190, first_block=2)
    assert S.digest_bytes(out1) == hashlib.sha256(msg).digest()
    prog = S.program(chained=True)
    iv_limbs = []
    for x in S.IV:
        iv_limbs += [x & 0xffff, x >> 16]
    oprm, prm = O.default_params(1, 5, 3), Params(1, 5, 3)
    pv0, pv1 = S.chained_publics(out0, iv_limbs), S.chained_publics(out1, out0[:16])
    assert S.check_rows(prog, t0, pv0) == [] and S.check_rows(prog, t1, pv1) == []
    p0 = O.prove_shard_air(prog, t0, pv0, oprm)
    p1 = O.prove_shard_air(prog, t1, pv1, oprm)
    assert verify_shard_air(prog, p0, 7, S.WIDTH, pv0, prm) == (0, 0)
    assert verify_shard_air(prog, p1, 7, S.WIDTH, pv1, prm) == (0, 0)
    assert verify_shard_air(prog, p1, 7, S.WIDTH, S.chained_publics(out1, iv_limbs), prm)[0] == -6          # shard 1 does not start from the standard value
    assert verify_shard_air(prog, p0, 7, S.WIDTH, S.chained_publics(out1, iv_limbs), prm)[0] == -6
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

    def check(chain_, dg_, length=190):
        return L.zkhip_verify_sha256_sharded(buf.ctypes.data_as(_lib.u8p), stride, lens, 2, chain_.ctypes.data_as(_lib.u32p), 1, dg_.ctypes.data_as(_lib.u8p),
                                             length, C.byref(prm), C.byref(bad), C.byref(reason)), bad.value, reason.value
    assert check(chain, dg) == (0, 0, 0)
    assert check(chain, dg, 189)[0] == -6 and check(chain, dg, 300)[0] == -6          # the same chain claimed for another length: the last shard's padding values differ; another shard count
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


def test_the_compressed_chain_verifier_accepts_the_oracles_proof(oracle):
    """zkhip_verify_sha256_compressed / zkhip_sha256_compress_key_host with no device: a 400-byte message as two chained shards of 4 blocks
    (the second with one inactive block), both proven by the oracle, verified in-circuit by the oracle's proof of the restatement's machine
    (tests/recursion_air.py, air mode on the chained program) -- the library's host verifier takes it from (digest, length, chain, key)"""
    import recursion_air as R
    from zktls_amd.device import sha256_compress_key_host, verify_sha256_compressed
    O = oracle
    msg = bytes((13 * i + 1) & 0xff for i in range(400))
    iprm, prm, oprm = Params(1, 2, 1), Params(1, 20, 8), O.default_params(1, 20, 8)
    blocks, prog = S.pad(msg), S.program(chained=True)
    assert len(blocks) == 7 * 64
    chain, inner, pubs = [list(S.IV)], [], []
    for s in range(2):
        t, out = S.trace(blocks[256 * s:256 * (s + 1)], 4, chain_in=chain[s], message_len=len(msg), first_block=4 * s)
        pin = []
        for x in chain[s]:
            pin += [x & 0xffff, x >> 16]
        chain.append([int(out[2 * k] | (out[2 * k + 1] << 16)) for k in range(8)])
        pubs.append(S.chained_publics(out, pin))
        inner.append(O.prove_shard_air(prog, t, pubs[-1], O.default_params(1, 2, 1)).tobytes())
    digest = b"".join(w.to_bytes(4, "big") for w in chain[2])
    assert digest == hashlib.sha256(msg).digest()
    sh, mains, pres, progs, tabs, pv = R.machine(inner, 8, S.WIDTH, pubs, 2, 1, program=prog)
    lns = [m.shape[0].bit_length() - 1 for m in mains]
    vk = sha256_compress_key_host(len(msg), 2, iprm, prm)
    assert vk.tolist() == [int(x) for x in O.machine_setup(pres, lns, oprm)]
    proof = O.prove_machine_keyed(mains, pres, progs, tabs, pv, oprm)
    ch = np.array(chain, dtype=np.uint32)
    assert verify_sha256_compressed(proof, digest, len(msg), ch, 2, vk, iprm, prm) == (0, 0)
    assert verify_sha256_compressed(proof, hashlib.sha256(b"x").digest(), len(msg), ch, 2, vk, iprm, prm)[0] == -6
    assert verify_sha256_compressed(proof, digest, len(msg) + 1, ch, 2, vk, iprm, prm)[0] == -6
    bad = ch.copy()
    bad[1, 0] ^= 1                                                          # a chain that does not link up: other public values than the proof's
    assert verify_sha256_compressed(proof, digest, len(msg), bad, 2, vk, iprm, prm)[0] == -6
    assert verify_sha256_compressed(proof, digest, len(msg), ch, 2, (vk + 1) % S.P, iprm, prm)[0] == -6
    assert verify_sha256_compressed(proof, digest, len(msg) + 64 * 4, ch, 2, vk, iprm, prm)[0] != 0          # (another number of shards: another machine)
