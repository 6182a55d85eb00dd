"""The keyed machine on the GPU (proof version 11): zkhip_machine_setup commits the preprocessed traces once, zkhip_prove_machine_keyed
proves against the key.  Roots and proof bytes against the oracle; the SHA-256 chip with a range table whose values are preprocessed."""
import numpy as np
import pytest

import machines as M
from zktls_amd._lib import Params, ZkHipError
from zktls_amd.device import verify_machine_keyed

pytestmark = pytest.mark.gpu


def shape_of(traces, pre):
    return ([t.shape[0].bit_length() - 1 for t in traces], [t.shape[1] for t in traces], [0 if p is None else p.shape[1] for p in pre])


def on_device(ctx, traces, pre, lns, ws, pws):
    chips = [(ctx.from_numpy(t), ln, w) for t, ln, w in zip(traces, lns, ws)]
    pre_chips = [(None if p is None else ctx.from_numpy(p), ln, pw) for p, ln, pw in zip(pre, lns, pws)]
    return chips, pre_chips


@pytest.mark.parametrize("shape", [(1, 6, 4), (2, 5, 0), (3, 4, 2)])
@pytest.mark.parametrize("size", [(6, 3), (9, 4), (7, 5)])
def test_byte_machine_bytes_equal_the_oracles(ctx, oracle, shape, size):
    O = oracle
    traces, pre, progs, tables, pub = M.byte_machine(*size)
    lns, ws, pws = shape_of(traces, pre)
    chips, pre_chips = on_device(ctx, traces, pre, lns, ws, pws)
    key = ctx.machine_setup(pre_chips, Params(*shape))
    assert (key.root == O.machine_setup(pre, lns, O.default_params(*shape))).all()
    proof = ctx.prove_machine_keyed(key, chips, progs, tables, pub, Params(*shape))
    assert proof.tobytes() == O.prove_machine_keyed(traces, pre, progs, tables, pub, O.default_params(*shape)).tobytes()
    assert verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, Params(*shape)) == (0, 0)
    # the key serves any number of proofs: another execution against the same table
    t2, pre2, _, _, _ = M.byte_machine(*size, seed=2)
    assert all((a is None and b is None) or (a == b).all() for a, b in zip(pre, pre2))
    chips2, _ = on_device(ctx, t2, pre, lns, ws, pws)
    proof2 = ctx.prove_machine_keyed(key, chips2, progs, tables, pub, Params(*shape))
    assert proof2.tobytes() == O.prove_machine_keyed(t2, pre, progs, tables, pub, O.default_params(*shape)).tobytes()
    key.close()


@pytest.mark.parametrize("seed", range(10))
def test_random_keyed_machines_bytes_equal_the_oracles(ctx, oracle, seed):
    O = oracle
    traces, pre, progs, tables, pub = M.random_keyed_machine(300 + seed)
    lns, ws, pws = shape_of(traces, pre)
    shape = [(1, 7, 3), (2, 5, 0), (3, 4, 2)][seed % 3]
    chips, pre_chips = on_device(ctx, traces, pre, lns, ws, pws)
    key = ctx.machine_setup(pre_chips, Params(*shape))
    assert (key.root == O.machine_setup(pre, lns, O.default_params(*shape))).all()
    proof = ctx.prove_machine_keyed(key, chips, progs, tables, pub, Params(*shape))
    assert proof.tobytes() == O.prove_machine_keyed(traces, pre, progs, tables, pub, O.default_params(*shape)).tobytes()
    assert verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, Params(*shape)) == (0, 0)
    assert O.verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, O.default_params(*shape)) == 0


def test_key_and_machine_must_agree(ctx, oracle):
    traces, pre, progs, tables, pub = M.byte_machine(6, 3)
    lns, ws, pws = shape_of(traces, pre)
    chips, pre_chips = on_device(ctx, traces, pre, lns, ws, pws)
    key = ctx.machine_setup(pre_chips, Params(1, 6, 4))
    with pytest.raises(ZkHipError):
        ctx.prove_machine_keyed(key, chips, progs, tables, pub, Params(2, 6, 4))           # another blowup than the key's
    with pytest.raises(ZkHipError):
        ctx.prove_machine_keyed(key, chips[:1], progs[:1], tables[:1], pub, Params(1, 6, 4))
    from zktls_amd import device as D
    other = D.Context(0)
    try:
        with pytest.raises(ZkHipError):
            other.prove_machine_keyed(key, chips, progs, tables, pub, Params(1, 6, 4))      # a key belongs to its context
    finally:
        other.close()
    with pytest.raises(ZkHipError):
        ctx.machine_setup([(None, ln, 0) for ln in lns], Params(1, 6, 4))                   # nothing to commit


def test_sha256_chip_with_a_preprocessed_range_table(ctx, oracle):
    """the SHA-256 compression chip sends four 16-bit limbs to a 2^16-row range table whose VALUES are preprocessed (fixed by the key, no
    counter constraints) and whose multiplicities are main columns; bytes equal the oracle's"""
    import hashlib
    import sha256_air as S
    from zktls_amd.device import sha256_air
    O = oracle
    V = O.air_var
    msg = bytes(range(251)) * 3
    d_sha, limbs = ctx.sha256_gen_trace(S.pad(msg))
    sha_pub = limbs.tolist()
    assert S.digest_bytes(sha_pub) == hashlib.sha256(msg).digest()
    sent = [S.OUT + 6, S.OUT + 7, S.OUT + 14, S.OUT + 15]
    sha_tab = O.interaction_table([(O.SEND, None, 16, [c]) for c in sent])
    values = np.zeros((1 << 16, 4), dtype=np.uint32)
    values[:, 0] = np.arange(1 << 16)
    d_counts = ctx.range_table(d_sha, 640, 1 << 10, sent, 16)                 # main columns (v, multiplicity, 0, 0)
    # combined row of the table: [v 0 0 0 | v m 0 0]; the program: a harmless first-row identity, the key fixes the values
    table_prog = O.air_program(8, S.N_PUBLIC, [(O.SEL_FIRST, [(1, [V(0)])])])
    table_tab = O.interaction_table([(O.RECEIVE, 5, 16, [0])])
    progs, tables = [table_prog, sha256_air()], [table_tab, sha_tab]
    prm, oprm = Params(1, 12, 4), O.default_params(1, 12, 4)
    key = ctx.machine_setup([(ctx.from_numpy(values), 16, 4), (None, 10, 0)], prm)
    proof = ctx.prove_machine_keyed(key, [(d_counts, 16, 4), (d_sha, 10, 640)], progs, tables, sha_pub, prm)
    host = [d_counts.download().reshape(-1, 4), d_sha.download().reshape(-1, 640)]
    assert proof.tobytes() == O.prove_machine_keyed(host, [values, None], progs, tables, sha_pub, oprm).tobytes()
    assert verify_machine_keyed(proof, [16, 10], [4, 640], [4, 0], key.root, progs, tables, sha_pub, prm) == (0, 0)
    wrong = list(sha_pub)
    wrong[0] ^= 1
    assert verify_machine_keyed(proof, [16, 10], [4, 640], [4, 0], key.root, progs, tables, wrong, prm)[0] == -6


import hashlib
import json
import os

KAT = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_kat.json")))


@pytest.mark.parametrize("name", sorted(KAT["keyed_machine_proofs"]))
def test_golden_keyed_machine_proofs_on_gpu(ctx, name):
    """the committed keys (roots) and proofs (sizes, SHA-256 of the bytes) reproduced by the HIP path without the oracle in the loop"""
    g = KAT["keyed_machine_proofs"][name]
    a = g["machine"]
    tr, pre, pg, tb, pub = M.byte_machine(*a[1:]) if a[0] == "byte" else M.random_keyed_machine(a[1])
    lns, ws, pws = shape_of(tr, pre)
    chips, pre_chips = on_device(ctx, tr, pre, lns, ws, pws)
    key = ctx.machine_setup(pre_chips, Params(*g["params"]))
    assert key.root.tolist() == g["root"]
    proof = ctx.prove_machine_keyed(key, chips, pg, tb, pub, Params(*g["params"]))
    assert proof.size == g["bytes"] and hashlib.sha256(proof.tobytes()).hexdigest() == g["sha256"]


@pytest.mark.parametrize("n_bytes", [0, 55, 753, 13217])
def test_sha256_machine_setup_prove_verify(ctx, oracle, n_bytes):
    """setup -> prove -> verify, the reference's three calls (sp1.rs:113, :116, :120), on the SHA-256 machine: the key and the proof bytes
    against the oracle proving the same machine rebuilt in Python, the digest against hashlib"""
    from zktls_amd.device import verify_sha256_machine
    O = oracle
    msg = np.random.default_rng(n_bytes).integers(0, 256, n_bytes, dtype=np.uint8).tobytes()
    prm, oprm = Params(1, 10, 4), O.default_params(1, 10, 4)
    key = ctx.sha256_setup(prm)
    digest, proof = ctx.prove_sha256_machine(key, msg, prm)
    assert digest == hashlib.sha256(msg).digest()
    tr, pre, pg, tb, pub = M.sha256_machine(msg)
    lns, ws, pws = shape_of(tr, pre)
    assert key.root.tolist() == O.machine_setup(pre, lns, oprm).tolist()
    assert proof.tobytes() == O.prove_machine_keyed(tr, pre, pg, tb, pub, oprm).tobytes()
    assert verify_sha256_machine(proof, digest, key.root, prm, len(msg)) == (0, 0)
    assert verify_sha256_machine(proof, digest, key.root, prm, len(msg) + 1)[0] == -6          # the digest is right, the stated length is not
    assert O.verify_machine_keyed(proof, lns, ws, pws, key.root, pg, tb, pub, oprm) == 0
    other = bytearray(digest)
    other[5] ^= 1
    assert verify_sha256_machine(proof, bytes(other), key.root, prm, len(msg))[0] == -6
    vk2 = key.root.copy()
    vk2[0] = (int(vk2[0]) + 1) % 2013265921
    assert verify_sha256_machine(proof, digest, vk2, prm, len(msg)) == (-6, 3)
    # the same key proves the next message
    digest2, proof2 = ctx.prove_sha256_machine(key, msg + b"x", prm)
    assert digest2 == hashlib.sha256(msg + b"x").digest() and verify_sha256_machine(proof2, digest2, key.root, prm, len(msg) + 1) == (0, 0)


def test_sha256_machine_above_the_table_height(ctx):
    """2^17 rows: the chip is taller than its 2^16-row table and comes first; accepted by the host verifier, digest against hashlib"""
    from zktls_amd.device import verify_sha256_machine
    msg = np.random.default_rng(5).integers(0, 256, (128 << 10) - 9, dtype=np.uint8).tobytes()
    prm = Params(1, 16, 4)
    key = ctx.sha256_setup(prm)
    digest, proof = ctx.prove_sha256_machine(key, msg, prm)
    assert digest == hashlib.sha256(msg).digest()
    w = np.frombuffer(proof.tobytes(), dtype=np.uint32)
    assert list(w[8:18]) == [17, 640, 1, 4, 0, 16, 4, 1, 1, 4]
    assert verify_sha256_machine(proof, digest, key.root, prm, len(msg)) == (0, 0)


def test_a_batch_of_transcripts_in_one_call(ctx, oracle):
    """BASELINE configs[2] in its honest form: 64 independent transcripts (the recorded 13 217-byte input with a counter appended), each
    proven as the keyed SHA-256 machine inside ONE library call; every digest against hashlib, every proof accepted under the one vk,
    three of them byte-equal to the oracle's proof of the same machine; a second call reuses the pooled contexts and their keys"""
    import time
    from zktls_amd.device import prove_transcripts, verify_sha256_machine
    base = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference", "guest_input0.cbor"), "rb").read()
    msgs = [base + i.to_bytes(4, "little") for i in range(64)]
    prm, oprm = Params(1, 20, 8), oracle.default_params(1, 20, 8)
    vk, res = prove_transcripts(msgs, prm, devices=[0], in_flight=4)
    assert len(res) == 64
    for m, (digest, proof) in zip(msgs, res):
        assert digest == hashlib.sha256(m).digest()
        assert verify_sha256_machine(proof, digest, vk, prm, len(m)) == (0, 0)
    for i in (0, 31, 63):
        tr, pre, pg, tb, pub = M.sha256_machine(msgs[i])
        assert res[i][1].tobytes() == oracle.prove_machine_keyed(tr, pre, pg, tb, pub, oprm).tobytes()
        if i == 0:
            assert vk.tolist() == oracle.machine_setup(pre, [t.shape[0].bit_length() - 1 for t in tr], oprm).tolist()
    t0 = time.perf_counter()
    vk2, res2 = prove_transcripts(msgs, prm, devices=[0], in_flight=4, verify=True)       # each proof checked inside the call, as sp1.rs:120
    dt = time.perf_counter() - t0
    assert vk2.tolist() == vk.tolist() and all(a[1].tobytes() == b[1].tobytes() for a, b in zip(res, res2))
    print("64 transcripts, second call: %.1f ms" % (dt * 1e3))
    # a failing job (a message beyond 1 MiB) is reported by index; the others are proven
    from zktls_amd._lib import ZkHipError
    with pytest.raises(ZkHipError):
        prove_transcripts([b"ok", bytes((1 << 20) + 1), b"fine"], prm, devices=[0], in_flight=2)


def test_a_key_of_tables_only_serves_machines_of_other_shapes(ctx, oracle):
    """zkhip_prove_machine_keyed_at: the key holds ONE entry (the byte table); machines whose user chip is shorter or taller than the table
    -- so that the table is the first chip or the second -- use it through an explicit assignment; bytes against the oracle either way"""
    O = oracle
    prm, oprm = Params(1, 6, 4), O.default_params(1, 6, 4)
    _, pre7, _, _, _ = M.byte_machine(7, 3)
    table_pre = [p for p in pre7 if p is not None][0]
    key = ctx.machine_setup([(ctx.from_numpy(table_pre), 6, 4)], prm)             # one entry: the 2^6-row table's preprocessed columns
    for log_users in (5, 6, 9):
        traces, pre, progs, tables, pub = M.byte_machine(log_users, 3)
        lns, ws, pws = shape_of(traces, pre)
        entries = [-1 if p is None else 0 for p in pre]
        chips = [(ctx.from_numpy(t), ln, w) for t, ln, w in zip(traces, lns, ws)]
        assert key.root.tolist() == O.machine_setup(pre, lns, oprm).tolist()      # the commitment does not depend on the other chips
        proof = ctx.prove_machine_keyed(key, chips, progs, tables, pub, prm, key_entries=entries)
        assert proof.tobytes() == O.prove_machine_keyed(traces, pre, progs, tables, pub, oprm).tobytes(), log_users
        assert verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, prm) == (0, 0)
    # an assignment that leaves the table out, uses it twice, or gives it to a chip of another height
    traces, pre, progs, tables, pub = M.byte_machine(9, 3)
    lns, ws, pws = shape_of(traces, pre)
    chips = [(ctx.from_numpy(t), ln, w) for t, ln, w in zip(traces, lns, ws)]
    for bad in ([-1, -1], [0, 0], [0, -1], [3, -1]):
        with pytest.raises(ZkHipError):
            ctx.prove_machine_keyed(key, chips, progs, tables, pub, prm, key_entries=bad)
