"""The SHA-256 compression chip on the GPU: on-device trace generation against the test-side restatement (cell for cell), digests
against hashlib, proof bytes against the oracle's generic constraint-program prover on the same trace, the product's and the
oracle's verifiers, and the independent pure-Python verifier."""
import hashlib

import numpy as np
import pytest

import pyverify
import sha256_air as S
from zktls_amd._lib import Params, ZkHipError
from zktls_amd.device import sha256_air, verify_sha256

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,total", [(0, None), (3, None), (56, None), (150, None), (500, None), (3, 4), (700, 16)])
def test_device_trace_equals_the_restatement(ctx, n, total):
    msg = bytes((11 * i + 3) & 0xff for i in range(n))
    t, pub = S.trace(S.pad(msg), total)
    d, limbs = ctx.sha256_gen_trace(S.pad(msg), total)
    got = d.download().reshape(-1, S.WIDTH)
    assert got.shape == t.shape and (got == t).all()
    assert limbs.tolist() == pub and S.digest_bytes(limbs.tolist()) == hashlib.sha256(msg).digest()


# (log_blowup, queries, pow_bits, logup_pairs, log_fold, log_final, hash_width)
@pytest.mark.parametrize("n,shape", [(3, (1, 10, 4)), (150, (1, 8, 0)), (150, (2, 6, 0, 0, 2, 2, 24)), (1000, (2, 5, 3, 0, 4, 2, 24))])
def test_prove_sha256_bytes_equal_the_oracles(ctx, oracle, n, shape):
    O = oracle
    msg = bytes((5 * i + 9) & 0xff for i in range(n))
    digest, proof = ctx.prove_sha256(msg, Params(*shape))
    assert digest == hashlib.sha256(msg).digest()
    prog = S.program()
    t, pub = S.trace(S.pad(msg))
    log_n = t.shape[0].bit_length() - 1
    oproof = O.prove_shard_air(prog, t, pub, O.default_params(*shape))
    assert proof.tobytes() == oproof.tobytes()
    assert verify_sha256(proof, digest, Params(*shape), len(msg)) == (0, 0)
    assert verify_sha256(proof, digest, Params(*shape), len(msg) + 1)[0] == -6        # the same digest under another stated length
    assert O.verify_shard_air(prog, proof, log_n, S.WIDTH, pub, O.default_params(*shape)) == 0
    if n <= 150:
        assert pyverify.verify(proof.tobytes(), log_n, S.WIDTH, pub, *shape, air=sha256_air()) is True
    wrong = bytearray(digest)
    wrong[31] ^= 0x80
    assert verify_sha256(proof, bytes(wrong), Params(*shape), len(msg))[0] == -6


def test_a_megabyte_transcript(ctx):
    """1 MiB - 9 bytes: 2^14 blocks, 2^20 rows x 640 columns (2.4 GiB trace, 4.9 GiB LDE): digest against hashlib, proof verified"""
    msg = np.random.default_rng(7).integers(0, 256, (1 << 20) - 9, dtype=np.uint8).tobytes()
    digest, proof = ctx.prove_sha256(msg, Params(1, 100, 16))
    assert digest == hashlib.sha256(msg).digest()
    assert verify_sha256(proof, digest, Params(1, 100, 16), len(msg)) == (0, 0)


def test_misuse_fails_loudly(ctx):
    with pytest.raises(ZkHipError):
        ctx.prove_sha256(b"abc", Params(1, 10, 4, 1))            # no lookup argument with a constraint program
    with pytest.raises(ZkHipError):
        ctx.sha256_gen_trace(S.pad(b"abc"), 3)                    # block count must be a power of two


def test_sixty_four_transcripts_in_one_call(ctx):
    """BASELINE configs[2] with a REAL statement per transcript: 64 distinct 13 KB inputs, each a SHA-256 chip trace generated on the
    device, proven by ONE zkhip_prove_shards_air_multi call (four in flight per device), every digest against hashlib, every proof
    accepted by the host verifier and bound to its own digest"""
    from zktls_amd.device import prove_shards_air_multi
    import os
    base = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference", "guest_input0.cbor"), "rb").read()
    prm = Params(1, 40, 8)
    traces, pubs, digests = [], [], []
    for i in range(64):
        msg = base + i.to_bytes(4, "little")
        d, limbs = ctx.sha256_gen_trace(S.pad(msg))
        traces.append(d)
        pubs.append(limbs.tolist())
        digests.append(hashlib.sha256(msg).digest())
        assert S.digest_bytes(pubs[-1]) == digests[-1]
    ctx.sync()
    proofs = prove_shards_air_multi(sha256_air(), traces, 14, 640, pubs, prm, devices=[0], in_flight=4)
    assert len(proofs) == 64 and len({p.tobytes() for p in proofs}) == 64
    for i, p in enumerate(proofs):
        assert verify_sha256(p, digests[i], prm, len(base) + 4) == (0, 0)
    assert verify_sha256(proofs[3], digests[4], prm, len(base) + 4)[0] == -6


def test_a_64_kib_message_bytes_equal_the_oracles(ctx, oracle):
    """2^10 blocks -> 2^16 rows x 640 columns: the program kernel, the LDE and the openings at a size where every kernel runs many
    workgroups; the oracle proves the same trace on all host cores"""
    import os
    O = oracle
    prev = min(8, os.cpu_count() or 1)
    O.set_threads(min(os.cpu_count() or 1, 96))
    try:
        msg = np.random.default_rng(3).integers(0, 256, (64 << 10) - 9, dtype=np.uint8).tobytes()
        digest, proof = ctx.prove_sha256(msg, Params(1, 30, 8))
        assert digest == hashlib.sha256(msg).digest()
        t, pub = S.trace(S.pad(msg))
        assert proof.tobytes() == O.prove_shard_air(S.program(), t, pub, O.default_params(1, 30, 8)).tobytes()
    finally:
        O.set_threads(prev)


@pytest.mark.parametrize("kib", [256, 1024])
def test_large_messages_bytes_equal_the_oracles(ctx, oracle, kib):
    """2^18 and 2^20 rows x 640 (the chip at the headline height): the oracle proves the trace the DEVICE generated (its cells are checked against the restatement at the
    smaller sizes above; the pure-Python generator would take minutes here)"""
    import os
    from zktls_amd.device import sha256_pad
    O = oracle
    prev = min(8, os.cpu_count() or 1)
    O.set_threads(min(os.cpu_count() or 1, 96))
    try:
        msg = np.random.default_rng(4).integers(0, 256, (kib << 10) - 9, dtype=np.uint8).tobytes()
        digest, proof = ctx.prove_sha256(msg, Params(1, 20, 4))
        assert digest == hashlib.sha256(msg).digest()
        d, limbs = ctx.sha256_gen_trace(sha256_pad(msg))
        t = d.download().reshape(-1, S.WIDTH)
        d.free()
        assert proof.tobytes() == O.prove_shard_air(S.program(), t, limbs.tolist(), O.default_params(1, 20, 4)).tobytes()
    finally:
        O.set_threads(prev)


# ---- a message of any length as a chain of shard proofs (zkhip_prove_sha256_sharded)
def test_sharded_proofs_bytes_equal_the_oracles(ctx, oracle):
    """1 000 bytes in shards of 4 blocks: 16 blocks with padding -> 4 shards; every shard's proof equals the oracle's proof of the Python
    restatement's trace from the same chaining value; the chain verifies; digest against hashlib"""
    from zktls_amd.device import prove_sha256_sharded, verify_sha256_sharded
    O = oracle
    msg = bytes((7 * i + 3) & 0xff for i in range(1000))
    prm, oprm = Params(1, 8, 4), O.default_params(1, 8, 4)
    res = prove_sha256_sharded(msg, 2, prm, devices=[0], in_flight=2)
    assert res.digest == hashlib.sha256(msg).digest() and len(res.proofs) == 4
    assert (res.chain[0] == np.array(S.IV, dtype=np.uint32)).all()
    assert verify_sha256_sharded(res, params=prm) == (0, 0, 0)
    blocks = S.pad(msg)
    prog = S.program(chained=True)
    for s, proof in enumerate(res.proofs):
        t, out = S.trace(blocks[256 * s:256 * (s + 1)], chain_in=[int(v) for v in res.chain[s]], message_len=len(msg), first_block=4 * s)
        pin = []
        for x in res.chain[s]:
            pin += [int(x) & 0xffff, int(x) >> 16]
        assert [int(out[2 * k] | (out[2 * k + 1] << 16)) for k in range(8)] == [int(v) for v in res.chain[s + 1]]
        assert proof.tobytes() == O.prove_shard_air(prog, t, S.chained_publics(out, pin), oprm).tobytes(), s
    # another digest, a swapped pair of shards, a shard from another message
    assert verify_sha256_sharded(res, digest=hashlib.sha256(b"other").digest(), params=prm)[0] == -6
    assert verify_sha256_sharded(res, params=prm, message_len=len(msg) - 1)[0] == -6 and verify_sha256_sharded(res, params=prm, message_len=len(msg) + 64)[0] == -6      # another stated length
    swapped = res.buf.copy()
    swapped[:res.stride], swapped[res.stride:2 * res.stride] = res.buf[res.stride:2 * res.stride], res.buf[:res.stride]
    assert verify_sha256_sharded(res, params=prm, proofs=swapped)[:2] == (-6, 0)
    res2 = prove_sha256_sharded(msg[:-1] + b"!", 2, prm, devices=[0], in_flight=2)
    mixed = res.buf.copy()
    mixed[3 * res.stride:] = res2.buf[3 * res.stride:]
    assert verify_sha256_sharded(res, params=prm, proofs=mixed)[:2] == (-6, 3)


def test_a_three_megabyte_body_as_a_chain_of_shards(ctx):
    """BASELINE configs[3] with a real statement: a 3 MiB body = 49 153 blocks -> three shards of 2^14 blocks (2^20 rows x 640 each) and a
    one-block shard; digest against hashlib, the chain accepted by the host verifier; a corrupted shard is named"""
    from zktls_amd.device import prove_sha256_sharded, verify_sha256_sharded
    msg = np.random.default_rng(9).integers(0, 256, 3 << 20, dtype=np.uint8).tobytes()
    prm = Params(1, 30, 8)
    res = prove_sha256_sharded(msg, 14, prm, devices=[0], in_flight=2)
    assert res.digest == hashlib.sha256(msg).digest() and len(res.proofs) == 4
    heights = [int(np.frombuffer(p[8:12].tobytes(), dtype=np.uint32)[0]) for p in res.proofs]
    assert heights == [20, 20, 20, 6]
    assert verify_sha256_sharded(res, params=prm) == (0, 0, 0)
    bad = res.buf.copy()
    bad[2 * res.stride + 5000] ^= 1
    assert verify_sha256_sharded(res, params=prm, proofs=bad)[:2] == (-6, 2)


# ---- the chain as ONE proof (zkhip_prove_sha256_compressed: the shards verified in-circuit, shard verifier machine in air mode)
def test_a_chain_of_shards_as_one_proof_bytes_equal_the_oracles(ctx, oracle):
    """messages of 1 000 bytes (16 padded blocks: four full shards of 4 blocks) and of 600 bytes (10 blocks: the third shard has two
    inactive blocks).  The ONE proof equals the oracle's generic keyed-machine proof of the restatement's arrays built from the shard
    proofs; its verifier takes digest, length, chain and the key (derived on the host)"""
    import recursion_air as R
    from zktls_amd.device import sha256_compress_key_host, verify_sha256_compressed, verify_shard_air
    O = oracle
    iprm, prm, oprm = Params(1, 4, 2), Params(1, 20, 8), O.default_params(1, 20, 8)
    for nbytes in (1000, 600):
        msg = bytes((11 * i + 5) & 0xff for i in range(nbytes))
        key = ctx.sha256_compress_setup(len(msg), 2, iprm, prm)
        assert sha256_compress_key_host(len(msg), 2, iprm, prm).tolist() == key.root.tolist()
        digest, chain, proof = ctx.prove_sha256_compressed(key, msg, 2, iprm, prm, devices=[0])
        assert digest == hashlib.sha256(msg).digest()
        assert verify_sha256_compressed(proof, digest, len(msg), chain, 2, key.root, iprm, prm) == (0, 0)
        # the restatement: the shard proofs (all 2^8 rows) made by the oracle from the restatement's traces, the machine over them
        blocks, prog = S.pad(msg), S.program(chained=True)
        n_shards = chain.shape[0] - 1
        assert n_shards == (len(blocks) // 64 + 3) // 4
        inner, pubs = [], []
        for s in range(n_shards):
            t, out = S.trace(blocks[256 * s:256 * (s + 1)], 4, chain_in=[int(v) for v in chain[s]], message_len=len(msg), first_block=4 * s)
            pin = []
            for x in chain[s]:
                pin += [int(x) & 0xffff, int(x) >> 16]
            pubs.append(S.chained_publics(out, pin))
            inner.append(O.prove_shard_air(prog, t, pubs[-1], O.default_params(1, 4, 2)).tobytes())
        sh, mains, pres, progs, tabs, pv = R.machine(inner, 8, S.WIDTH, pubs, 4, 2, program=prog)
        lns = [m.shape[0].bit_length() - 1 for m in mains]
        assert key.root.tolist() == O.machine_setup(pres, lns, oprm).tolist()
        assert proof.tobytes() == O.prove_machine_keyed(mains, pres, progs, tabs, pv, oprm).tobytes(), "the compressed chain differs from the oracle's proof"
        # refused: another digest, another length, a chain that does not link up, a chain that starts elsewhere
        assert verify_sha256_compressed(proof, hashlib.sha256(b"other").digest(), len(msg), chain, 2, key.root, iprm, prm)[0] == -6
        assert verify_sha256_compressed(proof, digest, len(msg) - 1, chain, 2, key.root, iprm, prm)[0] == -6
        bad = chain.copy()
        bad[1, 3] ^= 1
        assert verify_sha256_compressed(proof, digest, len(msg), bad, 2, key.root, iprm, prm)[0] == -6
        bad = chain.copy()
        bad[0, 0] ^= 1
        assert verify_sha256_compressed(proof, digest, len(msg), bad, 2, key.root, iprm, prm)[0] == -6
        key.close()


def test_a_three_megabyte_body_as_one_proof(ctx):
    """BASELINE configs[3] with a real statement, compressed: a 3 MiB body -> four shards of 2^20 rows x 640 (the last with one active block)
    -> ONE proof; checked from (digest, length, chain, key derived on the host)"""
    from zktls_amd.device import sha256_compress_key_host, verify_sha256_compressed
    msg = np.random.default_rng(9).integers(0, 256, 3 << 20, dtype=np.uint8).tobytes()
    iprm, prm = Params(1, 30, 8), Params(1, 30, 8)
    key = ctx.sha256_compress_setup(len(msg), 14, iprm, prm)
    digest, chain, proof = ctx.prove_sha256_compressed(key, msg, 14, iprm, prm, devices=[0])
    assert digest == hashlib.sha256(msg).digest() and chain.shape[0] == 5
    assert verify_sha256_compressed(proof, digest, len(msg), chain, 14, sha256_compress_key_host(len(msg), 14, iprm, prm), iprm, prm) == (0, 0)
    assert verify_sha256_compressed(proof, digest, len(msg) + 1, chain, 14, key.root, iprm, prm)[0] == -6
    print("3 MiB: one proof of %d bytes" % proof.size)
    key.close()
