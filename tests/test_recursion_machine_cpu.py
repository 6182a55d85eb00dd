"""MACHINE MODE of the shard verifier machine -- a verifier of version-11 keyed-machine proofs in-circuit (docs/RECURSION_NEXT.md),
restated in tests/recursion_machine.py -- on the CPU:
  * its programs hold row by row in plain integers on the witness of oracle-made keyed-machine proofs (one height, mixed heights either way,
    random machines with tuples of 1 .. 8 values and odd interaction counts), every bus balances, a flipped cell breaks a constraint or a bus;
  * THE TREE: two shard proofs -> two shard-verifier proofs (eight chips each, version 11) -> ONE proof that verifies both in-circuit; the
    oracle's generic keyed-machine prover proves it and two verifiers (the oracle's, the pure-Python multi-chip verifier) accept it from
    (the level-1 machine's description, the shard proofs' public values, the key) -- and refuse other public values."""
import numpy as np
import pytest

import machines as M
import recursion_air as R
import recursion_machine as RM
import sha256_air as S

SEED = 0x5A4B544C53


def inner(O, mains, pres, progs, tabs, pub, q=2, pb=1):
    """-> (the machine's description, its key, one proof of it)"""
    lns = [m.shape[0].bit_length() - 1 for m in mains]
    prm = O.default_params(1, q, pb)
    vk = [int(x) for x in O.machine_setup(pres, lns, prm)]
    proof = O.prove_machine_keyed(mains, pres, progs, tabs, pub, prm).tobytes()
    chips = [dict(ln=lns[c], W=mains[c].shape[1], Pw=0 if pres[c] is None else pres[c].shape[1], prog=progs[c], tab=tabs[c]) for c in range(len(mains))]
    return chips, vk, proof


def holds(sh, mains, pres, progs, tabs, pv):
    for name, main, pre, prog in zip(RM.order(sh), mains, pres, progs):
        rows = main if pre is None else np.concatenate([pre, main], axis=1)
        assert rows.shape[1] == int(prog[2]), name
        assert S.check_rows(prog, rows, pv) == [], name
    assert R.bus_balance(mains, pres, tabs) == []


@pytest.mark.parametrize("log_users,bits", [(6, 3), (7, 3), (5, 3)])          # one height; the chip without preprocessed columns taller; the table taller
def test_rows_hold_and_buses_balance_on_a_two_chip_machine(oracle, log_users, bits):
    mains, pres, progs, tabs, pub = M.byte_machine(log_users, bits)
    chips, vk, proof = inner(oracle, mains, pres, progs, tabs, pub)
    holds(*RM.machine(chips, vk, [proof], [pub], 2, 1))


@pytest.mark.parametrize("seed", [1, 4, 6])                                     # three heights; four chips, three of one height; tuples of several values, odd interaction counts
def test_rows_hold_and_buses_balance_on_random_machines(oracle, seed):
    mains, pres, progs, tabs, pub = M.random_keyed_machine(seed)
    chips, vk, proof = inner(oracle, mains, pres, progs, tabs, pub)
    holds(*RM.machine(chips, vk, [proof], [pub], 2, 1))


def test_a_join_of_two_machine_proofs_and_flipped_cells(oracle):
    """two proofs of one machine in ONE outer machine; then a flipped cell in every chip this mode adds or changes"""
    made = [M.byte_machine(7, 3, seed) for seed in (1, 2)]
    chips, vk, p0 = inner(oracle, *made[0])
    _, vk1, p1 = inner(oracle, *made[1])
    assert vk == vk1                                                            # (one key: the table's contents)
    sh, mains, pres, progs, tabs, pv = RM.machine(chips, vk, [p0, p1], [made[0][4], made[1][4]], 2, 1)
    holds(sh, mains, pres, progs, tabs, pv)
    names = RM.order(sh)
    rng = np.random.default_rng(7)
    used = {"P2R": sh.NP * sh.p2_rows, "ROWSUM": sh.NP * len(RM.rowsum_rows(sh)), "TS": sh.NP * sh.NTS, "QUERY": sh.NP * sh.Q * len(sh.hs), "OPENED": sh.NP * sh.NV // 2,
            "SCALARS": sh.NP * sh.C, "EVAL": sh.NP * len(sh.terms), "LOGUP": sh.NP * len(sh.lrows), "FOLD": sh.NP * sh.Q * sh.R}
    broken = tried = 0
    for name in ("ROWSUM", "OPENED", "QUERY", "LOGUP", "SCALARS", "EVAL", "FOLD", "TS", "P2R"):
        i = names.index(name)
        for _ in range(6):
            r, c = int(rng.integers(0, used[name])), int(rng.integers(0, mains[i].shape[1]))
            keep = int(mains[i][r, c])
            mains[i][r, c] = (keep + 1) % R.P
            rows = mains[i] if pres[i] is None else np.concatenate([pres[i], mains[i]], axis=1)
            bad = bool(S.check_rows(progs[i], rows, pv)) or bool(R.bus_balance(mains, pres, tabs))
            mains[i][r, c] = keep
            tried += 1
            broken += bad
    # (cells no constraint reads exist: padding columns, value slots whose receive flag is off, the boundary rows' unused sides)
    assert broken >= tried * 2 // 3, (broken, tried)


def test_the_tree_two_levels_proven_by_the_oracle(oracle):
    import pyverify_chips
    O = oracle
    log_n, width, q, pb = 5, 8, 1, 0
    prm1, prm2 = O.default_params(1, 1, 0), O.default_params(1, 2, 0)
    level1 = []
    for shard in range(2):
        pubs = [7, shard]
        proof0 = O.prove_shard(O.gen_trace(SEED, shard, log_n, width), pubs, O.default_params(1, q, pb)).tobytes()
        sh1, mains, pres, progs, tabs, pv = R.machine(proof0, log_n, width, pubs, q, pb)
        lns = [m.shape[0].bit_length() - 1 for m in mains]
        vk1 = O.machine_setup(pres, lns, prm1)
        level1.append((O.prove_machine_keyed(mains, pres, progs, tabs, pv, prm1).tobytes(), pv))
    chips = [dict(ln=lns[c], W=mains[c].shape[1], Pw=0 if pres[c] is None else pres[c].shape[1], prog=progs[c], tab=tabs[c]) for c in range(len(mains))]
    assert len(chips) == 8 and len({c["ln"] for c in chips}) == 2            # the shard verifier machine: eight chips of two heights, one of them without preprocessed columns
    sh, m2, p2, g2, t2, pv2 = RM.machine(chips, [int(x) for x in vk1], [o for o, _ in level1], [p for _, p in level1], 1, 0)
    holds(sh, m2, p2, g2, t2, pv2)
    lns2, w2, pw2 = [m.shape[0].bit_length() - 1 for m in m2], [m.shape[1] for m in m2], [0 if p is None else p.shape[1] for p in p2]
    vk2 = O.machine_setup(p2, lns2, prm2)
    top = O.prove_machine_keyed(m2, p2, g2, t2, pv2, prm2)
    assert pv2 == [7, 0, 7, 1]
    assert O.verify_machine_keyed(top, lns2, w2, pw2, vk2, g2, t2, pv2, prm2) == 0
    assert pyverify_chips.verify(top.tobytes(), lns2, w2, pv2, log_blowup=1, num_queries=2, pow_bits=0, programs=g2, tables=t2, pre_widths=pw2, pre_root=[int(x) for x in vk2]) is True
    assert O.verify_machine_keyed(top, lns2, w2, pw2, vk2, g2, t2, [7, 1, 7, 0], prm2) != 0          # the shards swapped: another statement
    # (the top's key is a function of the level-1 MACHINE -- its programs, tables, key -- and the number of proofs: RM.preprocessed takes the shape, no proof)


# ---------------------------------------------------------------------------------------------------------------- the product's machine == the restatement
def described(im, n_proofs=1):
    from zktls_amd.device import machine_verifier_describe
    out = []
    for i in range(10):
        p, ln, mw, pw = machine_verifier_describe(im, i, 0, n_proofs)
        t, _, _, _ = machine_verifier_describe(im, i, 1, n_proofs)
        e, _, _, _ = machine_verifier_describe(im, i, 2, n_proofs)
        out.append((p, t, e, ln, mw, pw))
    return out


def same_machine(chips, vk, q, pb, npub, n_proofs):
    from zktls_amd.device import InnerMachine
    sh = RM.MShape(chips, vk, q, pb, npub, n_proofs)
    RM.build_reads(sh)
    names, progs, tabs, pres, h = RM.order(sh), RM.programs(sh), RM.tables(sh), RM.preprocessed(sh), RM.heights(sh)
    got = described(InnerMachine(chips, vk, q, pb, npub), n_proofs)
    for nm, (p, t, e, ln, mw, pw) in zip(names, got):
        assert ln == h[nm] and np.array_equal(p, np.asarray(progs[nm], dtype=np.uint32)), (nm, "program")
        assert np.array_equal(t, np.asarray(tabs[nm], dtype=np.uint32)), (nm, "table")
        want = np.zeros(0, dtype=np.uint32) if pres[nm] is None else pres[nm].ravel()
        assert np.array_equal(e, want) and pw == (0 if pres[nm] is None else pres[nm].shape[1]), (nm, "preprocessed")
    return sh, names, pres, h


def chips_of(mains, pres, progs, tabs):
    lns = [m.shape[0].bit_length() - 1 for m in mains]
    return [dict(ln=lns[c], W=mains[c].shape[1], Pw=0 if pres[c] is None else pres[c].shape[1], prog=progs[c], tab=tabs[c]) for c in range(len(mains))], lns


def test_product_machine_equals_the_restatement_word_for_word(oracle):
    """csrc/machine_verifier.inl against tests/recursion_machine.py: programs, interaction tables and preprocessed traces of all ten chips, for
    two-chip machines (one height, mixed either way), random machines, the shard verifier's own eight-chip machine (THE TREE's level 2), joins;
    and the key computed on the host against the oracle's commitment to the restatement's preprocessed traces"""
    from zktls_amd._lib import Params
    from zktls_amd.device import InnerMachine, machine_verifier_key_host
    O = oracle
    cases = []
    for a, b in ((6, 3), (7, 3), (5, 3)):
        mains, pres, progs, tabs, pub = M.byte_machine(a, b)
        chips, lns = chips_of(mains, pres, progs, tabs)
        cases.append((chips, [int(x) for x in O.machine_setup(pres, lns, O.default_params(1, 2, 1))], 2, 1, len(pub)))
    for seed in (1, 4, 6):
        mains, pres, progs, tabs, pub = M.random_keyed_machine(seed)
        chips, lns = chips_of(mains, pres, progs, tabs)
        cases.append((chips, [int(x) for x in O.machine_setup(pres, lns, O.default_params(1, 3, 0))], 3, 0, len(pub)))
    sh1 = R.Shape(5, 8, 1, 0, 2)
    names1, progs1, tabs1, pres1, h1 = R.order(sh1), R.programs(sh1), R.tables(sh1), R.preprocessed(sh1), R.heights(sh1)
    chips1 = [dict(ln=h1[n], W=int(progs1[n][2]) - (0 if pres1[n] is None else pres1[n].shape[1]), Pw=0 if pres1[n] is None else pres1[n].shape[1], prog=progs1[n], tab=tabs1[n]) for n in names1]
    vk1 = [int(x) for x in O.machine_setup([pres1[n] for n in names1], [h1[n] for n in names1], O.default_params(1, 2, 0))]
    cases.append((chips1, vk1, 2, 0, 2))
    for i, (chips, vk, q, pb, npub) in enumerate(cases):
        for n_proofs in (1, 3) if i in (1, 6) else (1,):
            sh, names, pres, h = same_machine(chips, vk, q, pb, npub, n_proofs)
            if i in (0, 1, 4):
                pl, lns = [pres[n] for n in names], [h[n] for n in names]
                want = [int(x) for x in O.machine_setup(pl, lns, O.default_params(1, 20, 8))]
                assert machine_verifier_key_host(InnerMachine(chips, vk, q, pb, npub), Params(1, 20, 8), n_proofs).tolist() == want


def test_the_products_host_tables_equal_the_restatements_main_traces(oracle):
    """zkhip_prove_machine_verifier fills every main trace but the Poseidon2 chip's columns on the HOST (csrc/machine_verifier.inl, fill_proof: which IS the
    verification of the inner proof): those tables, word for word, against the restatement's main traces -- a join of two proofs of a mixed-height machine,
    a random machine, the shard verifier's own machine (the tree's upper level); the Poseidon2 rows as (input state, bit, index); a tampered proof refused"""
    import poseidon2_air as P2
    from zktls_amd.device import InnerMachine, machine_verifier_host_tables
    O = oracle
    cases = []
    made = [M.byte_machine(7, 3, seed) for seed in (1, 2)]
    chips, vk, p0 = inner(O, *made[0])
    cases.append((chips, vk, [p0, inner(O, *made[1])[2]], [made[0][4], made[1][4]], 2, 1))
    mains, pres, progs, tabs, pub = M.random_keyed_machine(6)
    chips, vk, pr = inner(O, mains, pres, progs, tabs, pub, 3, 0)
    cases.append((chips, vk, [pr], [pub], 3, 0))
    proof0 = O.prove_shard(O.gen_trace(SEED, 3, 5, 8), [7, 3], O.default_params(1, 1, 0)).tobytes()
    sh1, m1, p1, g1, t1, pv1 = R.machine(proof0, 5, 8, [7, 3], 1, 0)
    lns1 = [m.shape[0].bit_length() - 1 for m in m1]
    vk1 = [int(x) for x in O.machine_setup(p1, lns1, O.default_params(1, 2, 0))]
    chips1 = [dict(ln=lns1[c], W=m1[c].shape[1], Pw=0 if p1[c] is None else p1[c].shape[1], prog=g1[c], tab=t1[c]) for c in range(8)]
    cases.append((chips1, vk1, [O.prove_machine_keyed(m1, p1, g1, t1, pv1, O.default_params(1, 2, 0)).tobytes()], [pv1], 2, 0))
    for chips, vk, proofs, pubs, q, pb in cases:
        sh, mains, pres, progs, tabs, pv = RM.machine(chips, vk, proofs, pubs, q, pb)
        im = InnerMachine(chips, vk, q, pb, len(pubs[0]))
        blobs = [np.frombuffer(p, dtype=np.uint8) for p in proofs]
        for i, name in enumerate(RM.order(sh)):
            got = machine_verifier_host_tables(im, blobs, pubs, i)
            assert got is not None, name
            if name == "P2R":
                used = sh.NP * sh.p2_rows
                got = got.reshape(used, 18)
                want = mains[i][:used]
                assert np.array_equal(got[:, :16], want[:, P2.IN:P2.IN + 16]) and np.array_equal(got[:, 16], want[:, P2.BIT]) and np.array_equal(got[:, 17], want[:, RM.M_KP]), name
            else:
                assert np.array_equal(got, mains[i].ravel()), name
        bad = blobs[0].copy()
        bad[bad.size // 2] ^= 1
        assert machine_verifier_host_tables(im, [bad] + blobs[1:], pubs, 0) is None


def test_descriptions_the_machine_does_not_take_are_refused():
    """zkhip_machine_verifier_key_host on malformed or unsupported inner machines: an error, never a crash"""
    from zktls_amd._lib import Params, ZkHipError
    from zktls_amd.device import InnerMachine, machine_verifier_key_host
    import oracle_lib as O
    mains, pres, progs, tabs, pub = M.byte_machine(6, 3)
    chips, lns = chips_of(mains, pres, progs, tabs)
    vk = [1] * 8

    def refused(ch, q=2, pb=1, npub=1, n_proofs=1):
        with pytest.raises(ZkHipError):
            machine_verifier_key_host(InnerMachine(ch, vk, q, pb, npub), Params(1, 20, 8), n_proofs)
    assert machine_verifier_key_host(InnerMachine(chips, vk, 2, 1, 1), Params(1, 20, 8)).size == 8
    refused(chips[::-1] if chips[0]["ln"] != chips[1]["ln"] else [dict(chips[0], ln=5), dict(chips[1], ln=6)])          # not tallest first
    refused([dict(chips[0], W=chips[0]["W"] + 2)] + chips[1:])                                                            # a width that is no multiple of 4 (and not the program's)
    refused(chips, q=0)
    refused(chips, pb=29)
    refused(chips, npub=2)                                                                                               # another number of public values than the programs declare
    refused(chips, n_proofs=65)
    refused(chips * 9)                                                                                                   # more than 16 chips
    quintic = O.air_program(int(chips[0]["prog"][2]), 1, [(O.SEL_ALL, [(1, [O.air_var(0)] * 5)])])                       # log_quotient_degree 2
    refused([dict(chips[0], prog=quintic)] + chips[1:])
    nopre = [dict(c, Pw=0, prog=O.air_program(c["W"], 1, [(O.SEL_FIRST, [(1, [O.air_var(0)])])]), tab=O.interaction_table([(O.SEND, None, 5, [0])])) for c in chips]
    refused(nopre)                                                                                                       # no preprocessed columns at all: not a KEYED machine
    garbage = np.arange(40, dtype=np.uint32)
    refused([dict(chips[0], prog=garbage)] + chips[1:])
    refused([dict(chips[0], tab=garbage)] + chips[1:])


# ---------------------------------------------------------------------------------------------------------------- SP1's shard structure, core -> compress
SP1_SMALL = [(8, 24, 3, 1), (8, 32, 3, 0), (7, 16, 2, -1), (6, 32, 2, -1), (5, 8, 1, -1)]


def test_the_sp1_shaped_shard_is_joinable(oracle):
    """VERDICT r5 item 3 on the CPU: shards of SP1's structure -- chips of mixed heights, in-table LogUp pairs, a cross-table bus between two chips of one
    height, preprocessed columns -- as version-11 proofs of ONE keyed machine; the machine-mode machine over two of them holds row by row, its buses
    balance, the oracle proves it and accepts it from (the machine's description, the shards' public values, the key), and refuses the shards swapped"""
    O = oracle
    q, pb = 2, 1
    made = [M.sp1_shaped_machine(SP1_SMALL, seed=5, shard=s, pre=((3, 8),)) for s in (0, 1)]
    for mains, pres, progs, tabs, pub in made:                                  # the shards themselves: rows hold, buses balance (the cross pair only across its two tables)
        assert R.bus_balance(mains, pres, tabs) == []
        for m, p, g in zip(mains, pres, progs):
            assert S.check_rows(g, m if p is None else np.concatenate([p, m], axis=1), pub) == []
    lone = [made[0][0][0]], [None], [made[0][2][0]], [made[0][3][0]]
    assert R.bus_balance(lone[0], lone[1], lone[3]) != [], "a table of a cross pair balances on its own"
    chips, vk, p0 = inner(O, *made[0], q=q, pb=pb)
    _, vk1, p1 = inner(O, *made[1], q=q, pb=pb)
    assert vk == vk1                                                            # ONE key for every shard: the preprocessed columns are the program's, not the shard's
    pubs = [made[0][4], made[1][4]]
    sh, mains, pres, progs, tabs, pv = RM.machine(chips, vk, [p0, p1], pubs, q, pb)
    holds(sh, mains, pres, progs, tabs, pv)
    lns, ws, pws = [m.shape[0].bit_length() - 1 for m in mains], [m.shape[1] for m in mains], [0 if p is None else p.shape[1] for p in pres]
    prm = O.default_params(1, 3, 1)
    key = O.machine_setup(pres, lns, prm)
    top = O.prove_machine_keyed(mains, pres, progs, tabs, pv, prm)
    assert pv == pubs[0] + pubs[1]
    assert O.verify_machine_keyed(top, lns, ws, pws, key, progs, tabs, pv, prm) == 0
    assert O.verify_machine_keyed(top, lns, ws, pws, key, progs, tabs, pubs[1] + pubs[0], prm) != 0
    other = M.sp1_shaped_machine(SP1_SMALL, seed=5, shard=1, pre=((3, 8),), key_shard=77)
    _, vk2, p2 = inner(O, *other, q=q, pb=pb)
    assert vk2 != vk
    with pytest.raises(Exception):                                              # a shard proof of another key's machine: the restatement's own verification refuses it
        RM.machine(chips, vk, [p0, p2], pubs, q, pb)


def test_the_sp1_shaped_machine_of_the_library_is_the_tests(oracle):
    """zktls_amd.device.Sp1ShapedShard (what bench.py and the GPU tests prove) describes the machine tests/machines.py builds: programs, tables, widths"""
    from zktls_amd.device import Sp1ShapedShard
    s = Sp1ShapedShard(SP1_SMALL, ((3, 8),), 3)
    mains, pres, progs, tabs, pub = M.sp1_shaped_machine(SP1_SMALL, seed=5, shard=0, pre=((3, 8),))
    assert all(np.array_equal(a, b) for a, b in zip(s.programs, progs)) and all(np.array_equal(a, b) for a, b in zip(s.tables, tabs))
    assert s.widths == [m.shape[1] for m in mains] and s.pre_widths == [0 if p is None else p.shape[1] for p in pres]
    full = Sp1ShapedShard()
    assert full.cells == sum(w << ln for ln, w in [(20, 96), (20, 32), (19, 64), (18, 128), (16, 256), (14, 40)])      # bench.py's multichip shard
