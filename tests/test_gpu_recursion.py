"""The shard verifier as a machine on the GPU (csrc/shard_verifier.inl through the C ABI): the key of a shape, the outer proof of a real shard
proof -- bytes against the oracle's generic keyed-machine prover run on the Python restatement's arrays (tests/recursion_air.py) -- and the
verifier that is handed the shape, the inner proof's public values and the key: no byte of the inner proof."""
import numpy as np
import pytest

from zktls_amd._lib import Params, ZkHipError
from zktls_amd.device import shard_verifier_describe, verify_shard, verify_shard_recursive

pytestmark = pytest.mark.gpu
SEED = 0x5A4B544C53


@pytest.mark.parametrize("log_n,width,q,pb,pubs", [(5, 8, 4, 3, [1, 2, 3]), (6, 16, 5, 0, []), (9, 64, 12, 5, [7, 8, 9, 10, 11, 12, 13, 14, 15]), (21, 64, 6, 4, [9, 8]), (5, 8, 1, 0, []), (5, 1024, 2, 1, list(range(100, 164)))])      # (a shard above 2^20 rows; the smallest shape; the widest with the most public values)
def test_key_and_proof_bytes_equal_the_oracles(ctx, oracle, log_n, width, q, pb, pubs):
    import recursion_air as R
    O = oracle
    iprm, oprm, prm = Params(1, q, pb), O.default_params(1, 20, 8), Params(1, 20, 8)
    trace = ctx.gen_trace(SEED, 3, log_n, width)
    inner = ctx.prove_shard(trace, log_n, width, pubs, iprm)
    assert verify_shard(inner, log_n, width, pubs, iprm)[0] == 0
    key = ctx.shard_verifier_setup(log_n, width, q, pb, len(pubs), prm)
    sh, mains, pres, progs, tabs, pv = R.machine(inner.tobytes(), log_n, width, pubs, q, pb)
    lns = [m.shape[0].bit_length() - 1 for m in mains]
    assert key.root.tolist() == O.machine_setup(pres, lns, oprm).tolist(), "the key differs from the oracle's commitment to the restatement's preprocessed traces"
    from zktls_amd.device import shard_verifier_key_host
    assert shard_verifier_key_host(log_n, width, q, pb, len(pubs), prm).tolist() == key.root.tolist(), "the key computed on the host's cores (no device) differs from the device's"
    outer = ctx.prove_shard_verifier(key, inner, log_n, width, pubs, iprm, prm)
    assert outer.tobytes() == O.prove_machine_keyed(mains, pres, progs, tabs, pv, oprm).tobytes(), "outer proof bytes differ from the oracle's"
    assert verify_shard_recursive(outer, log_n, width, q, pb, pubs, key.root, prm) == (0, 0)
    assert O.verify_machine_keyed(outer, lns, [m.shape[1] for m in mains], [0 if p is None else p.shape[1] for p in pres], key.root, progs, tabs, pv, oprm) == 0
    if pubs:
        other = list(pubs)
        other[-1] += 1
        assert verify_shard_recursive(outer, log_n, width, q, pb, other, key.root, prm)[0] != 0
    # a second shard proof of the shape under the SAME key
    inner2 = ctx.prove_shard(ctx.gen_trace(SEED, 4, log_n, width), log_n, width, pubs, iprm)
    outer2 = ctx.prove_shard_verifier(key, inner2, log_n, width, pubs, iprm, prm)
    assert outer2.tobytes() != outer.tobytes() and verify_shard_recursive(outer2, log_n, width, q, pb, pubs, key.root, prm) == (0, 0)
    key.close()


def test_a_tampered_inner_proof_is_refused_by_the_prover(ctx):
    log_n, width, q, pb, pubs = 6, 16, 5, 2, [4, 5]
    iprm, prm = Params(1, q, pb), Params(1, 20, 8)
    inner = ctx.prove_shard(ctx.gen_trace(SEED, 9, log_n, width), log_n, width, pubs, iprm)
    key = ctx.shard_verifier_setup(log_n, width, q, pb, len(pubs), prm)
    for at in (40, 200, inner.size // 2, inner.size - 8):
        bad = inner.copy()
        bad[at] ^= 1
        with pytest.raises(ZkHipError) as e:
            ctx.prove_shard_verifier(key, bad, log_n, width, pubs, iprm, prm)
        assert e.value.code == -6, (at, str(e.value))          # ZKHIP_ERR_VERIFY whoever notices: the host pass (transcript, AIR identity, folds) or -- for a
        #                                                         digest inside a Merkle path, which only the device hashes -- the roots the P2R rows kernel arrives at
    with pytest.raises(ZkHipError):
        ctx.prove_shard_verifier(key, inner, log_n, width, [4, 6], iprm, prm)          # other public values than the proof's
    key.close()


def test_headline_shard_proof_verified_in_circuit(ctx, oracle):
    """the 2^20 x 256 shard proof of BASELINE configs[1] (100 queries, 16 proof-of-work bits): its whole verification in ONE outer proof; the
    oracle's verifier and the host verifier accept it with the machine as the library describes it, the shape's key and the public values"""
    O = oracle
    log_n, width, q, pb = 20, 256, 100, 16
    pubs = [11, 12, 13, 14, 15, 16, 17, 18, 19]
    iprm, prm, oprm = Params(1, q, pb), Params(1, 100, 16), O.default_params(1, 100, 16)
    trace = ctx.gen_trace(SEED, 0, log_n, width)
    inner = ctx.prove_shard(trace, log_n, width, pubs, iprm)
    trace.free()
    key = ctx.shard_verifier_setup(log_n, width, q, pb, len(pubs), prm)
    outer = ctx.prove_shard_verifier(key, inner, log_n, width, pubs, iprm, prm)
    assert ctx.prove_shard_verifier(key, inner, log_n, width, pubs, iprm, prm).tobytes() == outer.tobytes()
    assert verify_shard_recursive(outer, log_n, width, q, pb, pubs, key.root, prm) == (0, 0)
    progs, tabs, lns, widths, pws = [], [], [], [], []
    for i in range(8):
        p, ln, mw, pw = shard_verifier_describe(log_n, width, q, pb, len(pubs), i, 0)
        t, _, _, _ = shard_verifier_describe(log_n, width, q, pb, len(pubs), i, 1)
        progs.append(p), tabs.append(t), lns.append(ln), widths.append(mw), pws.append(pw)
    assert O.verify_machine_keyed(outer, lns, widths, pws, key.root, progs, tabs, pubs, oprm) == 0
    assert verify_shard_recursive(outer, log_n, width, q, pb, pubs[:-1] + [0], key.root, prm)[0] != 0
    print("headline: inner %d bytes, outer %d bytes, P2R 2^%d rows" % (inner.size, outer.size, lns[0]))
    key.close()


@pytest.mark.parametrize("nproofs,oshape", [(2, (1, 20, 8)), (3, (1, 20, 8)), (2, (2, 10, 4))])      # (the last: the outer proof at blowup 4, SP1's compress shape in small)
def test_the_join_bytes_equal_the_oracles(ctx, oracle, nproofs, oshape):
    """ONE outer proof for several inner proofs of one shape: key and bytes against the oracle on the restatement's arrays"""
    import recursion_air as R
    O = oracle
    log_n, width, q, pb = 6, 16, 5, 2
    iprm, oprm, prm = Params(1, q, pb), O.default_params(*oshape), Params(*oshape)
    pubs = [[3, 4, 50 + p] for p in range(nproofs)]
    inner = [ctx.prove_shard(ctx.gen_trace(SEED, 20 + p, log_n, width), log_n, width, pubs[p], iprm) for p in range(nproofs)]
    key = ctx.shard_verifier_setup(log_n, width, q, pb, 3, prm, n_proofs=nproofs)
    sh, mains, pres, progs, tabs, pv = R.machine([x.tobytes() for x in inner], log_n, width, pubs, q, pb)
    lns = [m.shape[0].bit_length() - 1 for m in mains]
    assert key.root.tolist() == O.machine_setup(pres, lns, oprm).tolist()
    outer = ctx.prove_shard_verifier(key, inner, log_n, width, pubs, iprm, prm)
    assert outer.tobytes() == O.prove_machine_keyed(mains, pres, progs, tabs, pv, oprm).tobytes(), "joined proof bytes differ from the oracle's"
    assert verify_shard_recursive(outer, log_n, width, q, pb, pv, key.root, prm, n_proofs=nproofs) == (0, 0)
    assert verify_shard_recursive(outer, log_n, width, q, pb, pubs[1] + pubs[0] + [v for p in pubs[2:] for v in p], key.root, prm, n_proofs=nproofs)[0] != 0
    # the proofs in another order are another statement: a new outer proof, accepted with the public values in THAT order
    outer2 = ctx.prove_shard_verifier(key, inner[::-1], log_n, width, pubs[::-1], iprm, prm)
    assert verify_shard_recursive(outer2, log_n, width, q, pb, [v for p in pubs[::-1] for v in p], key.root, prm, n_proofs=nproofs) == (0, 0)
    # one bad inner proof spoils the call
    bad = inner[0].copy()
    bad[bad.size // 3] ^= 1
    with pytest.raises(ZkHipError):
        ctx.prove_shard_verifier(key, [bad] + inner[1:], log_n, width, pubs, iprm, prm)
    key.close()


def test_sixteen_headline_shard_proofs_become_one_proof(ctx, oracle):
    """BASELINE configs[1] x 16: sixteen 2^20 x 256 shard proofs (15 MB) verified by ONE outer proof of about a megabyte; the oracle's verifier and
    the host verifier accept it with the 16 x 9 public values, the shape's key and nothing else"""
    O = oracle
    log_n, width, q, pb, n = 20, 256, 100, 16, 16
    iprm, prm, oprm = Params(1, q, pb), Params(1, 100, 16), O.default_params(1, 100, 16)
    pubs = [[1, 2, 3, 4, 5, 6, 7, 8, 100 + p] for p in range(n)]
    inner = []
    for p in range(n):
        tr = ctx.gen_trace(SEED, 200 + p, log_n, width)
        inner.append(ctx.prove_shard(tr, log_n, width, pubs[p], iprm))
        tr.free()
    key = ctx.shard_verifier_setup(log_n, width, q, pb, 9, prm, n_proofs=n)
    outer = ctx.prove_shard_verifier(key, inner, log_n, width, pubs, iprm, prm)
    flat = [v for p in pubs for v in p]
    assert verify_shard_recursive(outer, log_n, width, q, pb, flat, key.root, prm, n_proofs=n) == (0, 0)
    assert outer.size * 8 < sum(x.size for x in inner), "the join does not compress"
    progs, tabs, lns, widths, pws = [], [], [], [], []
    for i in range(8):
        p_, ln, mw, pw = shard_verifier_describe(log_n, width, q, pb, 9, i, 0, n)
        t_, _, _, _ = shard_verifier_describe(log_n, width, q, pb, 9, i, 1, n)
        progs.append(p_), tabs.append(t_), lns.append(ln), widths.append(mw), pws.append(pw)
    assert O.verify_machine_keyed(outer, lns, widths, pws, key.root, progs, tabs, flat, oprm) == 0
    flat[-1] += 1
    assert verify_shard_recursive(outer, log_n, width, q, pb, flat, key.root, prm, n_proofs=n)[0] != 0
    print("join: %d inner proofs, %d bytes -> %d bytes, P2R 2^%d rows" % (n, sum(x.size for x in inner), outer.size, lns[0]))
    key.close()


def test_the_largest_join_a_poseidon2_chip_of_2_pow_22_rows(ctx):
    """136 headline shard proofs (the most ONE join takes: zkhip_shard_verifier_max_proofs) under an outer proof at blowup 2; another outer blowup
    holds the Poseidon2 chip to 2^21 rows and refuses the shape at setup"""
    from zktls_amd.device import shard_verifier_max_proofs
    log_n, width, q, pb = 20, 256, 100, 16
    iprm, prm = Params(1, q, pb), Params(1, 100, 16)
    n = shard_verifier_max_proofs(log_n, width, q, pb, 9, prm)
    assert n == 136 and shard_verifier_max_proofs(log_n, width, q, pb, 9, Params(2, 50, 16)) == 68
    pubs = [[1, 2, 3, 4, 5, 6, 7, 8, p % 3] for p in range(n)]
    tr = ctx.gen_trace(SEED, 300, log_n, width)
    three = [ctx.prove_shard(tr, log_n, width, pubs[p], iprm) for p in range(3)]      # (three distinct proofs, repeated: the join does not care)
    tr.free()
    inner = [three[p % 3] for p in range(n)]
    with pytest.raises(ZkHipError):
        ctx.shard_verifier_setup(log_n, width, q, pb, 9, Params(2, 50, 16), n_proofs=n)
    key = ctx.shard_verifier_setup(log_n, width, q, pb, 9, prm, n_proofs=n)
    outer = ctx.prove_shard_verifier(key, inner, log_n, width, pubs, iprm, prm)
    flat = [v for p in pubs for v in p]
    assert verify_shard_recursive(outer, log_n, width, q, pb, flat, key.root, prm, n_proofs=n) == (0, 0)
    flat[-1] ^= 1
    assert verify_shard_recursive(outer, log_n, width, q, pb, flat, key.root, prm, n_proofs=n)[0] != 0
    assert outer.size * 60 < sum(x.size for x in inner)
    key.close()


# ---------------------------------------------------------------------------------------------------------------- air mode: version-7 inner proofs
def _sha_statement(digest, message_len):
    """the 91 public values of "digest = SHA-256 of a message of message_len bytes": what the verifier of a joined proof derives by itself"""
    from zktls_amd.device import sha256_padding_publics
    limbs = []
    for i in range(8):
        w = int.from_bytes(digest[4 * i:4 * i + 4], "big")
        limbs += [w & 0xffff, w >> 16]
    return limbs + sha256_padding_publics(message_len).tolist()


@pytest.mark.parametrize("kind,log_n,width,q,pb,nproofs", [("synthetic", 5, 8, 4, 3, 1), ("counter", 6, 16, 3, 2, 2), ("counter", 5, 8, 2, 0, 3), ("synthetic", 9, 64, 6, 4, 1)])
def test_air_mode_key_and_proof_bytes_equal_the_oracles(ctx, oracle, kind, log_n, width, q, pb, nproofs):
    """inner proofs of a constraint PROGRAM (version 7: zkhip_prove_shard_air) verified in-circuit: the machine of (shape, program) with the EVAL
    chip -- key and outer proof bytes against the oracle on the restatement's arrays; one proof and joins"""
    import recursion_air as R
    from zktls_amd.device import shard_verifier_key_host, verify_shard_air
    O = oracle
    iprm, oprm, prm = Params(1, q, pb), O.default_params(1, 20, 8), Params(1, 20, 8)
    inner, pubs = [], []
    for p in range(nproofs):
        if kind == "synthetic":
            pv_, prog = [4, 5, 6 + p], O.air_synthetic(width, 3)
            tr = ctx.gen_trace(SEED, 30 + p, log_n, width)
        else:
            prog = R.counter_program(width)
            t, pv_ = R.counter_trace(log_n, width, 500 + 9 * p, 70 + p, seed=p)
            tr = ctx.alloc(t.size)
            tr.upload(t)
        inner.append(ctx.prove_shard_air(prog, tr, log_n, width, pv_, iprm))
        assert verify_shard_air(prog, inner[-1], log_n, width, pv_, iprm)[0] == 0
        pubs.append(pv_)
    key = ctx.shard_verifier_setup(log_n, width, q, pb, 3, prm, n_proofs=nproofs, program=prog)
    sh, mains, pres, progs, tabs, pv = R.machine([x.tobytes() for x in inner], log_n, width, pubs, q, pb, program=prog)
    lns = [m.shape[0].bit_length() - 1 for m in mains]
    assert key.root.tolist() == O.machine_setup(pres, lns, oprm).tolist(), "the key differs from the oracle's commitment to the restatement's preprocessed traces"
    assert shard_verifier_key_host(log_n, width, q, pb, 3, prm, nproofs, program=prog).tolist() == key.root.tolist()
    outer = ctx.prove_shard_verifier(key, inner, log_n, width, pubs, iprm, prm, program=prog)
    assert outer.tobytes() == O.prove_machine_keyed(mains, pres, progs, tabs, pv, oprm).tobytes(), "outer proof bytes differ from the oracle's"
    assert verify_shard_recursive(outer, log_n, width, q, pb, pv, key.root, prm, n_proofs=nproofs, program=prog) == (0, 0)
    assert O.verify_machine_keyed(outer, lns, [m.shape[1] for m in mains], [0 if p is None else p.shape[1] for p in pres], key.root, progs, tabs, pv, oprm) == 0
    other = list(pv)
    other[-1] += 1
    assert verify_shard_recursive(outer, log_n, width, q, pb, other, key.root, prm, n_proofs=nproofs, program=prog)[0] != 0
    assert verify_shard_recursive(outer, log_n, width, q, pb, pv, key.root, prm, n_proofs=nproofs)[0] != 0              # the version-1 machine's verifier
    # the prover refuses: a flipped byte of an inner proof, other public values than the proof's, the key of another program
    bad = inner[0].copy()
    bad[bad.size // 2] ^= 1
    with pytest.raises(ZkHipError):
        ctx.prove_shard_verifier(key, [bad] + inner[1:], log_n, width, pubs, iprm, prm, program=prog)
    with pytest.raises(ZkHipError):
        ctx.prove_shard_verifier(key, inner, log_n, width, [pubs[0][:2] + [pubs[0][2] + 1]] + pubs[1:], iprm, prm, program=prog)
    if kind == "counter":
        prog2 = prog.copy()
        prog2[-4] = (int(prog2[-4]) + 1) % R.P
        with pytest.raises(ZkHipError):
            ctx.prove_shard_verifier(key, inner, log_n, width, pubs, iprm, prm, program=prog2)
    key.close()


def test_air_mode_a_sha256_proof_verified_in_circuit_bytes_equal_the_oracles(ctx, oracle):
    """zkhip_prove_sha256's proof of "digest = SHA-256 of a message of 100 bytes" (2^7 x 640, the chip's 815 constraints) -> an outer proof that a
    verifier checks from (the chip's program, digest, length, key).  Bytes against the oracle on the restatement's arrays."""
    import hashlib
    import recursion_air as R
    from zktls_amd.device import sha256_air, verify_sha256
    O = oracle
    q, pb = 3, 2
    iprm, oprm, prm = Params(1, q, pb), O.default_params(1, 20, 8), Params(1, 20, 8)
    msg = bytes((5 * i + 1) & 0xff for i in range(100))
    prog = sha256_air()
    digest, inner = ctx.prove_sha256(msg, iprm)
    assert digest == hashlib.sha256(msg).digest() and verify_sha256(inner, digest, iprm, len(msg)) == (0, 0)
    pubs = _sha_statement(digest, len(msg))
    log_n = int(np.frombuffer(inner[8:12].tobytes(), dtype=np.uint32)[0])
    assert log_n == 7                                                      # two blocks
    key = ctx.shard_verifier_setup(log_n, 640, q, pb, 91, prm, program=prog)
    sh, mains, pres, progs, tabs, pv = R.machine(inner.tobytes(), log_n, 640, pubs, q, pb, program=prog)
    lns = [m.shape[0].bit_length() - 1 for m in mains]
    assert key.root.tolist() == O.machine_setup(pres, lns, oprm).tolist()
    outer = ctx.prove_shard_verifier(key, inner, log_n, 640, pubs, iprm, prm, program=prog)
    assert outer.tobytes() == O.prove_machine_keyed(mains, pres, progs, tabs, pv, oprm).tobytes(), "outer proof bytes differ from the oracle's"
    assert verify_shard_recursive(outer, log_n, 640, q, pb, pubs, key.root, prm, program=prog) == (0, 0)
    assert verify_shard_recursive(outer, log_n, 640, q, pb, _sha_statement(hashlib.sha256(msg + b"x").digest(), len(msg)), key.root, prm, program=prog)[0] != 0
    assert verify_shard_recursive(outer, log_n, 640, q, pb, _sha_statement(digest, len(msg) + 1), key.root, prm, program=prog)[0] != 0
    key.close()


def test_air_mode_sixty_four_transcript_proofs_become_one_proof(ctx):
    """BASELINE configs[2]'s unit of work in small: 64 TLS-transcript-sized messages (13 221 bytes: 2^14 x 640 rows each), each proven by
    zkhip_prove_sha256 (20 queries), all 64 verified in-circuit by ONE outer proof.  Its verifier is handed the chip's program, the 64
    (digest, length) pairs and the key (derived on the host: no device)."""
    import hashlib
    from zktls_amd.device import sha256_air, shard_verifier_key_host, shard_verifier_max_proofs
    n, nbytes, q, pb = 64, 13221, 20, 8
    iprm, prm = Params(1, q, pb), Params(1, 20, 8)
    prog = sha256_air()
    msgs = [bytes((7 * i + 3 * p + 1) & 0xff for i in range(nbytes)) for p in range(n)]
    from zktls_amd.device import prove_transcripts
    inner, pubs = [], []
    for m, (d, pf) in zip(msgs, prove_transcripts(msgs, iprm, devices=[0], keyed=False)[1]):          # (zkhip_prove_transcripts_air: the batch, lock-step lanes)
        assert d == hashlib.sha256(m).digest()
        inner.append(pf), pubs.append(_sha_statement(d, len(m)))
    log_n = 14
    assert shard_verifier_max_proofs(log_n, 640, q, pb, 91, prm, program=prog) >= n
    key = ctx.shard_verifier_setup(log_n, 640, q, pb, 91, prm, n_proofs=n, program=prog)
    assert shard_verifier_key_host(log_n, 640, q, pb, 91, prm, n, program=prog).tolist() == key.root.tolist()
    outer = ctx.prove_shard_verifier(key, inner, log_n, 640, pubs, iprm, prm, program=prog)
    flat = [v for p in pubs for v in p]
    assert verify_shard_recursive(outer, log_n, 640, q, pb, flat, key.root, prm, n_proofs=n, program=prog) == (0, 0)
    total = sum(x.size for x in inner)
    assert outer.size * 4 < total, "the join does not compress"
    bad = list(flat)
    bad[91 * 17 + 2] ^= 1                                              # proof 17's digest
    assert verify_shard_recursive(outer, log_n, 640, q, pb, bad, key.root, prm, n_proofs=n, program=prog)[0] != 0
    bad = flat[:91 * 40] + _sha_statement(hashlib.sha256(msgs[40]).digest(), nbytes - 1) + flat[91 * 41:]      # proof 40: another length
    assert verify_shard_recursive(outer, log_n, 640, q, pb, bad, key.root, prm, n_proofs=n, program=prog)[0] != 0
    print("64 transcript proofs: %d bytes -> %d bytes" % (total, outer.size))
    key.close()


@pytest.mark.parametrize("in_flight", [1, 3])
def test_a_batch_of_joins_of_one_shape_is_the_joins_one_by_one(ctx, in_flight):
    """zkhip_prove_shard_verifier_batch: six shard proofs, two per join -> three joins dealt over pooled contexts (each makes the shape's key once
    and keeps it): the bytes and the key are those of zkhip_shard_verifier_setup / zkhip_prove_shard_verifier; a join with a tampered shard proof
    fails the call with the verifier's code; a second shape on the same pooled contexts gets its own key"""
    from zktls_amd.device import prove_shard_verifier_batch
    log_n, width, q, pb = 6, 16, 5, 2
    iprm, prm = Params(1, q, pb), Params(1, 20, 8)
    pubs = [[3, 4, 100 + s] for s in range(6)]
    inner = [ctx.prove_shard(ctx.gen_trace(SEED, 20 + s, log_n, width), log_n, width, pubs[s], iprm) for s in range(6)]
    key = ctx.shard_verifier_setup(log_n, width, q, pb, 3, prm, n_proofs=2)
    one_by_one = [ctx.prove_shard_verifier(key, inner[2 * j:2 * j + 2], log_n, width, pubs[2 * j:2 * j + 2], iprm, prm) for j in range(3)]
    for rep in range(2):                                                        # (the second call finds the keys with the pooled contexts)
        joins, vk = prove_shard_verifier_batch(inner, 2, log_n, width, pubs, iprm, prm, devices=[0], in_flight=in_flight, verify=True)
        assert vk.tolist() == key.root.tolist()
        assert [j.tobytes() for j in joins] == [j.tobytes() for j in one_by_one]
    for j in range(3):
        assert verify_shard_recursive(joins[j], log_n, width, q, pb, [v for p in pubs[2 * j:2 * j + 2] for v in p], vk, prm, n_proofs=2) == (0, 0)
    bad = [p.copy() for p in inner]
    bad[3][bad[3].size // 2] ^= 1
    with pytest.raises(ZkHipError) as e:
        prove_shard_verifier_batch(bad, 2, log_n, width, pubs, iprm, prm, devices=[0], in_flight=in_flight)
    assert e.value.code == -6
    with pytest.raises(ZkHipError):
        prove_shard_verifier_batch(inner[:5], 2, log_n, width, pubs[:5], iprm, prm, devices=[0])          # five proofs do not make joins of two
    # another shape (three per join, the outer proof at blowup 4) on the same contexts: another key
    joins3, vk3 = prove_shard_verifier_batch(inner, 3, log_n, width, pubs, iprm, Params(2, 10, 4), devices=[0], in_flight=in_flight, verify=True)
    key3 = ctx.shard_verifier_setup(log_n, width, q, pb, 3, Params(2, 10, 4), n_proofs=3)
    assert vk3.tolist() == key3.root.tolist() and vk3.tolist() != vk.tolist() and len(joins3) == 2
    assert joins3[1].tobytes() == ctx.prove_shard_verifier(key3, inner[3:], log_n, width, pubs[3:], iprm, Params(2, 10, 4)).tobytes()
    key.close(), key3.close()


def test_host_tables_kept_between_calls_start_from_zero(ctx):
    """the recursion provers keep their large host tables between calls (zeroed again on a thread of their own, csrc/shard_verifier.inl WordPool): a join
    made with tables that served ANOTHER set of shard proofs is the join made with fresh pages, byte for byte"""
    import time
    from zktls_amd import _lib
    log_n, width, q, pb, n = 12, 128, 100, 8, 16
    iprm, prm = Params(1, q, pb), Params(1, 20, 8)
    sets = []
    for k in range(2):
        pubs = [[k, 7, s] for s in range(n)]
        sets.append((pubs, [ctx.prove_shard(ctx.gen_trace(SEED, 700 + 100 * k + s, log_n, width), log_n, width, pubs[s], iprm) for s in range(n)]))
    key = ctx.shard_verifier_setup(log_n, width, q, pb, 3, prm, n_proofs=n)
    _lib.load().zkhip_release_cached_contexts()                              # (empties the pool: the first call below takes fresh pages)
    fresh = ctx.prove_shard_verifier(key, sets[0][1], log_n, width, sets[0][0], iprm, prm)
    time.sleep(0.5)                                                           # (the recycling thread zeroes and parks the tables)
    other = ctx.prove_shard_verifier(key, sets[1][1], log_n, width, sets[1][0], iprm, prm)
    time.sleep(0.5)
    again = ctx.prove_shard_verifier(key, sets[0][1], log_n, width, sets[0][0], iprm, prm)
    assert again.tobytes() == fresh.tobytes() and other.tobytes() != fresh.tobytes()
    for proof, (pubs, _) in ((again, sets[0]), (other, sets[1])):
        assert verify_shard_recursive(proof, log_n, width, q, pb, [v for p in pubs for v in p], key.root, prm, n_proofs=n) == (0, 0)
    key.close()
