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


# ---------------------------------------------------------------------------------------------------------------- buses are multisets: every receiver must be NAMED
#   What the audit (R.bus_ambiguity) is told about the machines, with the reason:
#   * BUS_E0 / BUS_E1 are lookups into a RELATION: a Poseidon2 leaf row proves "the pair (e0, e1) sits at index k of layer l's tree" by itself (its path rows
#     end in the layer's root, which TS ties to the transcript), so any such row may serve any fold row that claims that (tree, index, pair);
#   * on BUS_Q the position after the key names the query: in the TOP form (7 values) it is IDX, which the row also receives from SAMPLES under the
#     PREPROCESSED query number (BUS_I); in the LOWER-height form (8 values) it is IDX0, constant along the query's rows and equal to that IDX (two constraints
#     of the QUERY program, one of the FOLD program) -- the column 8ca4614 added.
import fri_air as F
FUNCTION_TABLES = (F.BUS_E0, F.BUS_E1)
NAMED = {(F.BUS_Q, 7): [1], (F.BUS_Q, 8): [1]}


def _table_entries(tab):
    t, p, out = [int(x) for x in tab], 3, []
    for _ in range(int(tab[1])):
        sign, mult, bus, nv = t[p:p + 4]
        out.append((sign, None if mult == 0xFFFFFFFF else mult, bus, t[p + 4:p + 4 + nv]))
        p += 4 + nv
    return out


def _mixed_height_join(O):
    """a machine-mode machine over 2 proofs x 3 queries of an inner machine with four heights (the SP1-shaped one): -> (sh, names, mains, pres, progs, tabs, pv)"""
    q, pb = 3, 1
    made = [M.sp1_shaped_machine(SP1_SMALL, seed=5, shard=s, pre=((3, 8),)) for s in (0, 1)]
    chips, vk, p0 = inner(O, *made[0], q=q, pb=pb)
    _, _, p1 = inner(O, *made[1], q=q, pb=pb)
    sh, mains, pres, progs, tabs, pv = RM.machine(chips, vk, [p0, p1], [made[0][4], made[1][4]], q, pb)
    return sh, RM.order(sh), mains, pres, progs, tabs, pv


def test_every_receiver_of_every_bus_is_named_in_all_three_machines(oracle):
    """the audit over honest traces of >= 2 proofs x >= 2 queries: the ten-chip machine (machine mode, mixed heights), the eight-chip one (version-1 shard proofs)
    and the nine-chip one (air mode) -- no two receivers of a bus agree on everything that names them and differ in what they take"""
    O = oracle
    sh, names, mains, pres, progs, tabs, pv = _mixed_height_join(O)
    assert len(sh.hs) >= 3 and sh.NP == 2 and sh.Q == 3
    assert R.bus_ambiguity(mains, pres, tabs, NAMED, FUNCTION_TABLES) == []
    found = {f[0] for f in R.bus_ambiguity(mains, pres, tabs)}                  # told nothing, it reports exactly the buses documented above
    assert found == {F.BUS_E0, F.BUS_E1, F.BUS_Q}
    log_n, width, q, pb = 6, 16, 3, 1
    pubs = [[5, 6, 30 + s] for s in range(2)]
    proofs = [O.prove_shard(O.gen_trace(SEED, 40 + s, log_n, width), pubs[s], O.default_params(1, q, pb)).tobytes() for s in range(2)]
    _, m8, p8, _, t8, _ = R.machine(proofs, log_n, width, pubs, q, pb)
    assert len(m8) == 8 and R.bus_ambiguity(m8, p8, t8, NAMED, FUNCTION_TABLES) == []
    prog = O.air_synthetic(width, 3)
    proofs7 = [O.prove_shard_air(prog, O.gen_trace(SEED, 40 + s, log_n, width), pubs[s], O.default_params(1, q, pb)).tobytes() for s in range(2)]
    _, m9, p9, _, t9, _ = R.machine(proofs7, log_n, width, pubs, q, pb, program=prog)
    assert len(m9) == 9 and R.bus_ambiguity(m9, p9, t9, NAMED, FUNCTION_TABLES) == []


def test_two_queries_cannot_exchange_their_lower_heights(oracle):
    """the regression 8ca4614 lacked.  Before it the fold chain handed a lower height's reduced opening over as (layer, index there, point, value) -- nothing
    in the tuple named the QUERY.  (a) the audit refuses that tuple: with IDX0 dropped from both sides of the lower-height form, two queries' receivers of one
    (proof, height) agree on the key and nothing else names them; (b) the exchange itself: the QUERY rows of two queries at one lower height take each
    other's (index, point) -- every constraint of the QUERY program but the IDX0 ones is recomputed and holds -- and the two fold chains take each other's
    reduced openings: BUS_Q balances under the old tuple and does NOT under the new one."""
    O = oracle
    sh, names, mains, pres, progs, tabs, pv = _mixed_height_join(O)
    qi, fi = names.index("QUERY"), names.index("FOLD")
    m = RM.query_cols()

    def old_table(tab):
        ent = []
        for sign, mult, bus, cols in _table_entries(tab):
            if bus == F.BUS_Q and len(cols) == 8:
                cols = cols[:1] + cols[2:]                                       # (key, IDX0, IDX, X, value) -> (key, IDX, X, value)
            ent.append((sign, mult, bus, cols))
        return O.interaction_table(ent)
    old_tabs = [old_table(t) if i in (qi, fi) else t for i, t in enumerate(tabs)]
    assert R.bus_balance(mains, pres, old_tabs) == []                           # (honest traces balance either way)
    # (a) IDX may be declared a name only where something pins it: the TOP rows (BUS_I).  The old lower-height receive shares its 7-value form with them, and its
    # IDX is pinned by nothing -- so nothing may be declared for it, and the audit finds the receivers of one (proof, height) indistinguishable
    assert F.BUS_Q in {f[0] for f in R.bus_ambiguity(mains, pres, old_tabs, {}, FUNCTION_TABLES)}
    # (b) the exchange
    ev = [e for e in R.bus_events(mains, pres, tabs)[F.BUS_Q] if len(e[5]) == 8]
    recv = {e[5]: e for e in ev if e[0] == R.RECV}
    sends = [e for e in ev if e[0] == R.SEND]
    pair = None
    for a in sends:
        for b in sends:
            if a[5][0] == b[5][0] and a[5][1] != b[5][1]:                        # one (proof, height), two queries
                pair = (a, b)
                break
        if pair:
            break
    assert pair, "the machine has no two queries at one lower height"
    (sa, sb), (ra, rb) = pair, (recv[pair[0][5]], recv[pair[1][5]])
    mq, mf = mains[qi].copy().astype(np.int64), mains[fi].copy().astype(np.int64)
    pw_q = pres[qi].shape[1]

    def col(name):
        return m[name] - pw_q
    rows_q = {"a": ra[3], "b": rb[3]}
    keep = {k: (int(mq[r, col("IDX")]), int(mq[r, col("XQ")])) for k, r in rows_q.items()}
    new_ro = {}
    for k, other in (("a", "b"), ("b", "a")):
        r = rows_q[k]
        idx, xq = keep[other]
        mq[r, col("IDX")], mq[r, col("XQ")] = idx, xq
        ext = lambda name: [int(v) for v in mq[r, col(name):col(name) + 4]]  # noqa: E731
        x = [RM.GEN * xq % R.P, 0, 0, 0]
        i1, i2 = RM.pyref.ext_inv(RM.e_sub(x, ext("ZETA"))), RM.pyref.ext_inv(RM.e_sub(x, ext("ZNX")))
        p1, p2 = RM.ext_mul(RM.e_sub(ext("AZ"), ext("YZ")), i1), RM.ext_mul(RM.e_sub(ext("AN"), ext("YN")), i2)
        ro = RM.e_add(p1, p2)
        for name, val in (("I1", i1), ("I2", i2), ("P1", p1), ("P2", p2), ("RO", ro)):
            mq[r, col(name):col(name) + 4] = val
        new_ro[k] = ro
    INJ, INJF = F.inj_cols(sh.R)
    mf[sa[3], INJ:INJ + 4], mf[sb[3], INJ:INJ + 4] = new_ro["b"], new_ro["a"]     # chain A takes what row B now holds under A's (index, point), and the other way round
    ex_mains = [mq.astype(np.uint32) if i == qi else (mf.astype(np.uint32) if i == fi else x) for i, x in enumerate(mains)]
    # the QUERY rows still satisfy their program -- IDX0 was not touched, so even the constraints 8ca4614 added hold
    rows = np.concatenate([pres[qi], ex_mains[qi]], axis=1)
    assert S.check_rows(progs[qi], rows, pv) == []

    def busq_unbalanced(tb):
        return [k for k in R.bus_balance(ex_mains, pres, tb) if k[0] == F.BUS_Q]
    assert busq_unbalanced(old_tabs) == [], "under the old tuple the exchange is invisible to the bus"
    assert busq_unbalanced(tabs) != [], "under the tuple with IDX0 the bus refuses the exchange"
