"""The machine on the GPU (zkhip_prove_machine, proof version 10): the generic permutation trace, the lookup constraints on the
quotient domain and whole proofs, byte for byte against the oracle."""
import numpy as np
import pytest

import machines as M
from zktls_amd._lib import Params
from zktls_amd.device import verify_machine

pytestmark = pytest.mark.gpu


def shape_of(traces):
    return [t.shape[0].bit_length() - 1 for t in traces], [t.shape[1] for t in traces]


@pytest.mark.parametrize("log_table,log_users,shape", [(5, 6, (1, 8, 4)), (6, 8, (2, 6, 0)), (8, 11, (1, 10, 6)), (10, 13, (3, 5, 2))])
def test_range_machine_bytes_equal_the_oracles(ctx, oracle, log_table, log_users, shape):
    O = oracle
    traces, progs, tables, pub = M.range_machine(log_table, log_users, seed=log_users)
    lns, ws = shape_of(traces)
    chips = [(ctx.from_numpy(t), ln, w) for t, ln, w in zip(traces, lns, ws)]
    proof = ctx.prove_machine(chips, progs, tables, pub, Params(*shape))
    oproof = O.prove_machine(traces, progs, tables, pub, O.default_params(*shape))
    assert proof.size == oproof.size
    assert proof.tobytes() == oproof.tobytes()
    assert verify_machine(proof, lns, ws, progs, tables, pub, Params(*shape)) == (0, 0)
    assert O.verify_machine(proof, lns, ws, progs, tables, pub, O.default_params(*shape)) == 0
    if log_users <= 8:
        import pyverify_chips
        assert pyverify_chips.verify(proof.tobytes(), lns, ws, pub, *shape, programs=progs, tables=tables) is True


def test_machine_with_synthetic_chips_and_an_odd_number_of_interactions(ctx, oracle):
    O = oracle
    traces, progs, tables, pub = M.range_machine(5, 7, seed=3)            # USER has three interactions: a pair column and a single one
    syn = O.gen_trace(5, 1, 7, 8)
    syn2 = O.gen_trace(5, 2, 4 + 1, 12)
    traces, progs, tables = [traces[0], traces[1], syn, traces[2], syn2], [progs[0], progs[1], None, progs[2], None], [tables[0], tables[1], None, tables[2], None]
    lns, ws = shape_of(traces)
    chips = [(ctx.from_numpy(t), ln, w) for t, ln, w in zip(traces, lns, ws)]
    proof = ctx.prove_machine(chips, progs, tables, pub, Params(1, 7, 3))
    assert proof.tobytes() == O.prove_machine(traces, progs, tables, pub, O.default_params(1, 7, 3)).tobytes()
    assert verify_machine(proof, lns, ws, progs, tables, pub, Params(1, 7, 3)) == (0, 0)


def test_a_prover_cannot_hide_an_unbalanced_lookup(ctx, oracle):
    O = oracle
    traces, progs, tables, pub = M.range_machine(5, 6)
    traces[2][1, 1] = (int(traces[2][1, 1]) + 1) % O.P
    lns, ws = shape_of(traces)
    chips = [(ctx.from_numpy(t), ln, w) for t, ln, w in zip(traces, lns, ws)]
    proof = ctx.prove_machine(chips, progs, tables, pub, Params(1, 6, 4))
    assert verify_machine(proof, lns, ws, progs, tables, pub, Params(1, 6, 4)) == (-6, 11)


def test_sha256_chip_with_its_limbs_range_checked_by_a_table(ctx, oracle):
    """a machine of two real tables: the SHA-256 compression chip sends the four 16-bit limbs its own constraints do not range-check
    (OUT of d and h) to a 2^16-row range table with multiplicities; bytes equal the oracle's, digest against hashlib"""
    import hashlib
    import sha256_air as S
    from zktls_amd.device import sha256_air
    O = oracle
    V = O.air_var
    msg = bytes(range(251)) * 3                                              # 753 bytes -> 12 blocks -> 16 blocks, 2^10 rows
    sha_t, sha_pub = S.trace(S.pad(msg))
    assert S.digest_bytes(sha_pub) == hashlib.sha256(msg).digest()
    d_sha, limbs = ctx.sha256_gen_trace(S.pad(msg))
    sent = [S.OUT + 6, S.OUT + 7, S.OUT + 14, S.OUT + 15]                    # OUT limb pairs of d (word 3) and h (word 7)
    sha_tab = O.interaction_table([(O.SEND, None, 16, [c]) for c in sent])
    table = np.zeros((1 << 16, 4), dtype=np.uint32)
    table[:, 0] = np.arange(1 << 16)
    table[:, 1] = np.bincount(sha_t[:, sent].ravel(), minlength=1 << 16)
    d_table = ctx.range_table(d_sha, 640, 1 << 10, sent, 16)                 # the same table, counted on the device
    assert (d_table.download().reshape(-1, 4) == table).all()
    table_prog = O.air_program(4, S.N_PUBLIC, [(O.SEL_FIRST, [(1, [V(0)])]),
                                       (O.SEL_TRANSITION, [(1, [V(0, True)]), (O.P - 1, [V(0)]), (O.P - 1, [])])])
    table_tab = O.interaction_table([(O.RECEIVE, 1, 16, [0])])
    progs, tables = [table_prog, sha256_air()], [table_tab, sha_tab]
    chips = [(d_table, 16, 4), (d_sha, 10, 640)]
    proof = ctx.prove_machine(chips, progs, tables, sha_pub, Params(1, 12, 4))
    oproof = O.prove_machine([table, sha_t], [table_prog, S.program()], tables, sha_pub, O.default_params(1, 12, 4))
    assert proof.tobytes() == oproof.tobytes()
    assert verify_machine(proof, [16, 10], [4, 640], progs, tables, sha_pub, Params(1, 12, 4)) == (0, 0)
    wrong = list(sha_pub)
    wrong[0] ^= 1
    assert verify_machine(proof, [16, 10], [4, 640], progs, tables, wrong, Params(1, 12, 4))[0] == -6


def test_range_table_refuses_values_it_does_not_hold(ctx):
    from zktls_amd._lib import ZkHipError
    t = ctx.from_numpy(np.array([[1, 2, 3, 40]], dtype=np.uint32).repeat(32, axis=0))
    assert (ctx.range_table(t, 4, 32, [0, 1, 2], 5).download().reshape(-1, 4)[:4, 1] == [0, 32, 32, 32]).all()
    with pytest.raises(ZkHipError):
        ctx.range_table(t, 4, 32, [3], 5)                                    # 40 >= 2^5


@pytest.mark.parametrize("seed", range(12))
def test_random_machines_bytes_equal_the_oracles(ctx, oracle, seed):
    O = oracle
    traces, progs, tables, pub = M.random_machine(100 + seed)
    lns, ws = shape_of(traces)
    shape = (1 + seed % 3, 6, 3)
    chips = [(ctx.from_numpy(t), ln, w) for t, ln, w in zip(traces, lns, ws)]
    proof = ctx.prove_machine(chips, progs, tables, pub, Params(*shape))
    assert proof.tobytes() == O.prove_machine(traces, progs, tables, pub, O.default_params(*shape)).tobytes()
    assert verify_machine(proof, lns, ws, progs, tables, pub, Params(*shape)) == (0, 0)


def test_a_machine_of_thirty_tables(ctx, oracle):
    """an SP1 shard holds two to three dozen chips: 30 tables, eight of them of one height, a range machine among synthetic tables"""
    O = oracle
    traces, progs, tables, pub = M.range_machine(6, 8, seed=9)               # heights 9, 8, 6
    extra = [(10, 8)] * 3 + [(9, 4)] * 6 + [(8, 12)] * 7 + [(7, 4)] * 8 + [(5, 8)] * 3          # 27 synthetic tables
    allt = [(t, p_, tb) for t, p_, tb in zip(traces, progs, tables)] + [(O.gen_trace(5, 70 + i, h, w), None, None) for i, (h, w) in enumerate(extra)]
    allt.sort(key=lambda e: -e[0].shape[0])
    traces, progs, tables = [e[0] for e in allt], [e[1] for e in allt], [e[2] for e in allt]
    lns, ws = shape_of(traces)
    assert len(traces) == 30 and max(lns.count(h) for h in lns) == 8
    chips = [(ctx.from_numpy(t), ln, w) for t, ln, w in zip(traces, lns, ws)]
    proof = ctx.prove_machine(chips, progs, tables, pub, Params(1, 8, 4))
    assert proof.tobytes() == O.prove_machine(traces, progs, tables, pub, O.default_params(1, 8, 4)).tobytes()
    assert verify_machine(proof, lns, ws, progs, tables, pub, Params(1, 8, 4)) == (0, 0)
    # one more table of the crowded height is refused
    from zktls_amd._lib import ZkHipError
    t9 = O.gen_trace(5, 99, 7, 4)
    i = lns.index(7)
    with pytest.raises(ZkHipError):
        ctx.prove_machine(chips[:i] + [(ctx.from_numpy(t9), 7, 4)] + chips[i:], progs[:i] + [None] + progs[i:], tables[:i] + [None] + tables[i:], pub, Params(1, 8, 4))


def test_sha256_machine_at_2_18_rows_bytes_equal_the_oracles(ctx, oracle):
    """the SHA-256 chip (2^18 rows x 640) + its 2^16-row range table, both generated / counted on the device: bytes against the oracle
    proving the downloaded tables on all host cores"""
    import hashlib
    import os
    import sha256_air as S
    from zktls_amd.device import sha256_air, sha256_pad
    O = oracle
    V = O.air_var
    prev = min(8, os.cpu_count() or 1)
    O.set_threads(min(os.cpu_count() or 1, 96))
    try:
        msg = np.random.default_rng(8).integers(0, 256, (256 << 10) - 9, dtype=np.uint8).tobytes()
        d_sha, limbs = ctx.sha256_gen_trace(sha256_pad(msg))
        pub = limbs.tolist()
        assert S.digest_bytes(pub) == hashlib.sha256(msg).digest()
        sent = [S.OUT + 6, S.OUT + 7, S.OUT + 14, S.OUT + 15]
        d_table = ctx.range_table(d_sha, 640, 1 << 18, sent, 16)
        sha_tab = O.interaction_table([(O.SEND, None, 16, [c]) for c in sent])
        table_prog = O.air_program(4, S.N_PUBLIC, [(O.SEL_FIRST, [(1, [V(0)])]), (O.SEL_TRANSITION, [(1, [V(0, True)]), (O.P - 1, [V(0)]), (O.P - 1, [])])])
        table_tab = O.interaction_table([(O.RECEIVE, 1, 16, [0])])
        progs, tables = [sha256_air(), table_prog], [sha_tab, table_tab]
        proof = ctx.prove_machine([(d_sha, 18, 640), (d_table, 16, 4)], progs, tables, pub, Params(1, 16, 4))
        host = [d_sha.download().reshape(-1, 640), d_table.download().reshape(-1, 4)]
        assert proof.tobytes() == O.prove_machine(host, progs, tables, pub, O.default_params(1, 16, 4)).tobytes()
        assert verify_machine(proof, [18, 16], [640, 4], progs, tables, pub, Params(1, 16, 4)) == (0, 0)
    finally:
        O.set_threads(prev)


import hashlib
import json
import os

KAT = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_kat.json")))


@pytest.mark.parametrize("name", sorted(KAT["machine_proofs"]))
def test_golden_machine_proofs_on_gpu(ctx, name):
    """the committed machine proofs (sizes and SHA-256 of the bytes) reproduced by the HIP prover without the oracle in the loop"""
    g = KAT["machine_proofs"][name]
    a = g["machine"]
    tr, pg, tb, pub = M.range_machine(*a[1:]) if a[0] == "range" else M.random_machine(a[1])
    lns, ws = shape_of(tr)
    proof = ctx.prove_machine([(ctx.from_numpy(t), ln, w) for t, ln, w in zip(tr, lns, ws)], pg, tb, pub, Params(*g["params"]))
    assert proof.size == g["bytes"] and hashlib.sha256(proof.tobytes()).hexdigest() == g["sha256"]
