"""The shard verifier as a machine (csrc/shard_verifier.inl, restated in tests/recursion_air.py), on the CPU:
  * the restatement's programs hold row by row in plain integers on the witness of a real (oracle-made) shard proof, every bus balances, and a
    flipped cell breaks a constraint or a bus;
  * the product's machine -- programs, interaction tables, preprocessed traces, all host code -- equals the restatement word for word;
  * the key (the commitment to the preprocessed traces) does not depend on the inner proof;
  * the oracle's generic keyed-machine prover proves the machine, and three verifiers (oracle, product host verifier through
    zkhip_verify_shard_recursive, the pure-Python multi-chip verifier) accept -- handed the shape, the inner proof's public values and the key,
    no byte of the inner proof -- and refuse other public values, another key, another shape."""
import numpy as np
import pytest

import recursion_air as R
import sha256_air as S

SEED = 0x5A4B544C53
SHAPES = [(5, 8, 4, 3, [1, 2, 3]), (6, 16, 5, 0, []), (5, 40, 3, 2, list(range(20, 31))), (5, 8, 1, 0, [])]      # (the last: the smallest shape the machine takes)


def inner_proof(O, log_n, width, q, pb, pubs, shard=0):
    return O.prove_shard(O.gen_trace(SEED, shard, log_n, width), pubs, O.default_params(1, q, pb)).tobytes()


@pytest.mark.parametrize("log_n,width,q,pb,pubs", SHAPES)
def test_programs_hold_and_buses_balance(oracle, log_n, width, q, pb, pubs):
    proof = inner_proof(oracle, log_n, width, q, pb, pubs)
    sh, mains, pres, progs, tabs, pv = R.machine(proof, log_n, width, pubs, q, pb)
    for name, main, pre, prog in zip(R.order(sh), mains, pres, progs):
        rows = main if pre is None else np.concatenate([pre, main], axis=1)
        assert rows.shape[1] == int(prog[2])
        assert S.check_rows(prog, rows, pv) == [], name
    assert R.bus_balance(mains, pres, tabs) == []


def test_a_flipped_cell_breaks_a_constraint_or_a_bus(oracle):
    log_n, width, q, pb, pubs = SHAPES[0]
    proof = inner_proof(oracle, log_n, width, q, pb, pubs)
    sh, mains, pres, progs, tabs, pv = R.machine(proof, log_n, width, pubs, q, pb)
    names = R.order(sh)
    rng = np.random.default_rng(5)
    tried = 0
    for name in names:
        i = names.index(name)
        used = {"P2R": sh.p2_rows, "ROWSUM": sh.Q * (sh.WB + 1), "FOLD": sh.Q * sh.R, "TS": sh.NTS, "QUERY": sh.Q, "OPENED": sh.G, "SAMPLES": sh.NS, "SCALARS": 1}[name]
        for _ in range(6):
            r, c = int(rng.integers(0, used)), int(rng.integers(0, mains[i].shape[1]))
            keep = int(mains[i][r, c])
            mains[i][r, c] = (keep + 1) % R.P
            rows = mains[i] if pres[i] is None else np.concatenate([pres[i], mains[i]], axis=1)
            broken = bool(S.check_rows(progs[i], rows, pv)) or bool(R.bus_balance(mains, pres, tabs))
            mains[i][r, c] = keep
            # (cells no constraint reads: unused padding columns of a chip, and the TS rows' copies of kept sponge words that are balanced by construction)
            unused = {"P2R": c > R.M_KP, "FOLD": c >= R.F.L_REC + sh.R, "QUERY": c >= R.query_cols().n - R.Q_PRE, "SCALARS": c >= R.scalars_cols(sh).n - R.SC_PRE,
                      "TS": (c >= 8 and c < 16 and r != 0) or (c >= 16 and not (r in (sh.TA, sh.TQ, sh.TF) or sh.TL0 <= r < sh.TP))}.get(name, False)
            assert broken or unused, (name, r, c)
            tried += 1
    assert tried == 48


def test_product_machine_equals_the_restatement_word_for_word():
    from zktls_amd.device import shard_verifier_describe
    for log_n, width, q, pb, npub, nproofs in ((5, 8, 4, 3, 3, 1), (6, 16, 5, 0, 0, 1), (7, 24, 9, 4, 9, 1), (5, 40, 3, 2, 11, 1), (8, 64, 12, 5, 2, 1),
                                               (5, 8, 4, 3, 3, 2), (5, 8, 4, 3, 3, 3), (6, 16, 5, 1, 9, 4), (21, 8, 2, 1, 1, 1), (22, 8, 1, 0, 0, 2), (5, 8, 4, 3, 3, 65), (5, 8, 1, 0, 0, 1), (5, 1024, 2, 1, 64, 1)):      # (the tallest shards the prover takes; a join of more than 64)
        sh = R.Shape(log_n, width, q, pb, npub, nproofs)
        names, progs, tabs, pres, h = R.order(sh), R.programs(sh), R.tables(sh), R.preprocessed(sh), R.heights(sh)
        for i, nm in enumerate(names):
            p, ln, mw, pw = shard_verifier_describe(log_n, width, q, pb, npub, i, 0, nproofs)
            t, _, _, _ = shard_verifier_describe(log_n, width, q, pb, npub, i, 1, nproofs)
            e, _, _, _ = shard_verifier_describe(log_n, width, q, pb, npub, i, 2, nproofs)
            assert ln == h[nm] and p.tolist() == [int(x) for x in progs[nm]], (nm, "program", nproofs)
            assert t.tolist() == [int(x) for x in tabs[nm]], (nm, "table", nproofs)
            want = np.zeros(0, dtype=np.uint32) if pres[nm] is None else pres[nm].ravel()
            assert e.tolist() == want.tolist() and pw == (0 if pres[nm] is None else pres[nm].shape[1]), (nm, "preprocessed", nproofs)


def test_host_key_equals_the_oracles_key(oracle):
    """zkhip_shard_verifier_key_host (csrc/host_key.cpp: coset LDE and mixed-height commitment on the host's cores, no device) against the
    oracle's machine_setup over the RESTATEMENT's preprocessed traces, at both outer blowups; the generic entry on the same tables"""
    from zktls_amd._lib import Params, to_monty
    from zktls_amd.device import shard_verifier_key_host, machine_key_host
    for log_n, width, q, pb, npub, nproofs in ((5, 8, 4, 3, 3, 1), (6, 16, 5, 0, 0, 1), (5, 8, 4, 3, 3, 2), (7, 24, 9, 4, 9, 3), (5, 40, 3, 2, 11, 1), (10, 64, 6, 2, 9, 4)):
        sh = R.Shape(log_n, width, q, pb, npub, nproofs)
        names, pres, h = R.order(sh), R.preprocessed(sh), R.heights(sh)
        pl, lns = [pres[n] for n in names], [h[n] for n in names]
        for blow, nq, pw in ((1, 20, 8), (2, 10, 4)):
            want = [int(x) for x in oracle.machine_setup(pl, lns, oracle.default_params(blow, nq, pw))]
            assert shard_verifier_key_host(log_n, width, q, pb, npub, Params(blow, nq, pw), nproofs).tolist() == want, (log_n, width, nproofs, blow)
        mont = [None if t is None else to_monty(t) for t in pl]
        assert machine_key_host(mont, lns, Params(1, 20, 8)).tolist() == [int(x) for x in oracle.machine_setup(pl, lns, oracle.default_params(1, 20, 8))]


def test_the_join_one_outer_proof_for_several_inner_proofs(oracle):
    """n_proofs inner proofs of one shape, ONE outer proof: programs hold, buses balance, the oracle proves it, three verifiers accept the public
    values of all proofs in order and refuse them swapped"""
    import pyverify_chips
    from zktls_amd._lib import Params
    from zktls_amd.device import verify_shard_recursive
    O = oracle
    log_n, width, q, pb = 5, 8, 4, 3
    oprm, prm = O.default_params(1, 20, 8), Params(1, 20, 8)
    for nproofs in (2, 3):
        pubs = [[1, 2, 10 + p] for p in range(nproofs)]
        proofs = [inner_proof(O, log_n, width, q, pb, pubs[p], shard=p) for p in range(nproofs)]
        sh, mains, pres, progs, tabs, pv = R.machine(proofs, log_n, width, pubs, q, pb)
        for name, main, pre, prog in zip(R.order(sh), mains, pres, progs):
            rows = main if pre is None else np.concatenate([pre, main], axis=1)
            assert S.check_rows(prog, rows, pv) == [], name
        assert R.bus_balance(mains, pres, tabs) == []
        lns = [m.shape[0].bit_length() - 1 for m in mains]
        widths, pws = [m.shape[1] for m in mains], [0 if p is None else p.shape[1] for p in pres]
        vk = O.machine_setup(pres, lns, oprm)
        op = O.prove_machine_keyed(mains, pres, progs, tabs, pv, oprm)
        assert op.size < 60 * sum(len(p) for p in proofs)              # (small shapes: the outer proof is bigger than these tiny inner proofs; at the headline shape it is 1 / 11 of sixteen)
        assert O.verify_machine_keyed(op, lns, widths, pws, vk, progs, tabs, pv, oprm) == 0
        assert verify_shard_recursive(op, log_n, width, q, pb, pv, vk, prm, n_proofs=nproofs) == (0, 0)
        assert pyverify_chips.verify(op.tobytes(), lns, widths, pv, log_blowup=1, num_queries=20, pow_bits=8, programs=progs, tables=tabs, pre_widths=pws, pre_root=[int(x) for x in vk]) is True
        swapped = pubs[1] + pubs[0] + [v for p in pubs[2:] for v in p]
        assert verify_shard_recursive(op, log_n, width, q, pb, swapped, vk, prm, n_proofs=nproofs)[0] != 0
        assert verify_shard_recursive(op, log_n, width, q, pb, pv, vk, prm, n_proofs=nproofs + 1)[0] != 0


def test_oracle_proves_the_machine_and_three_verifiers_take_no_byte_of_the_inner_proof(oracle):
    import pyverify_chips
    from zktls_amd._lib import Params
    from zktls_amd.device import verify_shard_recursive
    O = oracle
    log_n, width, q, pb, pubs = SHAPES[0]
    oprm, prm = O.default_params(1, 20, 8), Params(1, 20, 8)
    outers, roots = [], []
    for shard in (0, 1):
        proof = inner_proof(O, log_n, width, q, pb, pubs[:2] + [shard])
        sh, mains, pres, progs, tabs, pv = R.machine(proof, log_n, width, pubs[:2] + [shard], q, pb)
        lns = [m.shape[0].bit_length() - 1 for m in mains]
        roots.append(O.machine_setup(pres, lns, oprm))
        outers.append((O.prove_machine_keyed(mains, pres, progs, tabs, pv, oprm), pv, lns, [m.shape[1] for m in mains], [0 if p is None else p.shape[1] for p in pres], progs, tabs))
    assert roots[0].tolist() == roots[1].tolist(), "the key depends on the inner proof"
    vk = roots[0]
    for op, pv, lns, widths, pws, progs, tabs in outers:
        assert O.verify_machine_keyed(op, lns, widths, pws, vk, progs, tabs, pv, oprm) == 0
        assert verify_shard_recursive(op, log_n, width, q, pb, pv, vk, prm) == (0, 0)
        assert pyverify_chips.verify(op.tobytes(), lns, widths, pv, log_blowup=1, num_queries=20, pow_bits=8, programs=progs, tables=tabs, pre_widths=pws, pre_root=[int(x) for x in vk]) is True
    op, pv = outers[0][0], outers[0][1]
    assert verify_shard_recursive(op, log_n, width, q, pb, outers[1][1], vk, prm)[0] != 0                 # the other proof's public values
    assert verify_shard_recursive(op, log_n, width, q, pb, pv, (vk + 1) % R.P, prm)[0] != 0               # another key
    # another shape, where the shape shows in a program or a height (the number of queries of this small machine lives in the KEY alone: the key is
    # what binds a proof to its shape, and a verifier takes the key of the shape it means)
    assert verify_shard_recursive(op, log_n + 1, width, q, pb, pv, vk, prm)[0] != 0
    assert verify_shard_recursive(op, log_n, width + 8, q, pb, pv, vk, prm)[0] != 0
    assert verify_shard_recursive(op, log_n, width, q, pb + 1, pv, vk, prm)[0] != 0
    bad = op.copy()
    bad[len(bad) // 2] ^= 1
    assert verify_shard_recursive(bad, log_n, width, q, pb, pv, vk, prm)[0] != 0


# ---------------------------------------------------------------------------------------------------------------- air mode (version-7 inner proofs)
# The inner proofs are proofs of a constraint PROGRAM (zkhip_prove_shard_air, zkhip_prove_sha256 ...): the machine of (shape, program) has a ninth
# chip, EVAL, one row per term of that program (csrc/shard_verifier.inl "AIR MODE").
def air_inner(O, kind, log_n, width, q, pb, shard=0):
    """-> (program, proof bytes, public values)"""
    if kind == "synthetic":
        pubs = [4, 5, 6 + shard]
        prog = O.air_synthetic(width, len(pubs))
        t = O.gen_trace(SEED, shard, log_n, width)
    else:
        prog = R.counter_program(width)
        t, pubs = R.counter_trace(log_n, width, 1000 + 77 * shard, 9 + shard, seed=shard)
    return prog, O.prove_shard_air(prog, t, pubs, O.default_params(1, q, pb)).tobytes(), pubs


AIR_SHAPES = [("synthetic", 5, 8, 4, 3), ("counter", 5, 8, 3, 2), ("counter", 6, 16, 2, 0), ("synthetic", 5, 24, 1, 1)]


@pytest.mark.parametrize("kind,log_n,width,q,pb", AIR_SHAPES)
def test_air_mode_programs_hold_and_buses_balance(oracle, kind, log_n, width, q, pb):
    prog, proof, pubs = air_inner(oracle, kind, log_n, width, q, pb)
    sh, mains, pres, progs, tabs, pv = R.machine(proof, log_n, width, pubs, q, pb, program=prog)
    assert "EVAL" in R.order(sh)
    for name, main, pre, pr in zip(R.order(sh), mains, pres, progs):
        rows = main if pre is None else np.concatenate([pre, main], axis=1)
        assert rows.shape[1] == int(pr[2])
        assert S.check_rows(pr, rows, pv) == [], name
    assert R.bus_balance(mains, pres, tabs) == []


def test_air_mode_a_flipped_cell_breaks_a_constraint_or_a_bus(oracle):
    kind, log_n, width, q, pb = AIR_SHAPES[1]
    prog, proof, pubs = air_inner(oracle, kind, log_n, width, q, pb)
    sh, mains, pres, progs, tabs, pv = R.machine(proof, log_n, width, pubs, q, pb, program=prog)
    names = R.order(sh)
    rng = np.random.default_rng(11)
    for name in ("EVAL", "OPENED", "SCALARS", "TS"):
        i = names.index(name)
        used = {"TS": sh.NTS, "OPENED": sh.G, "SCALARS": 1, "EVAL": len(sh.terms)}[name]
        for _ in range(8):
            r, c = int(rng.integers(0, used)), int(rng.integers(0, mains[i].shape[1]))
            keep = int(mains[i][r, c])
            mains[i][r, c] = (keep + 1) % R.P
            rows = mains[i] if pres[i] is None else np.concatenate([pres[i], mains[i]], axis=1)
            broken = bool(S.check_rows(progs[i], rows, pv)) or bool(R.bus_balance(mains, pres, tabs))
            mains[i][r, c] = keep
            oc = R.opened_cols(R.op_pre(sh))
            unused = {"OPENED": c >= oc["YNO"] + 4 - R.op_pre(sh), "SCALARS": c >= R.scalars_cols(sh).n - R.sc_pre(sh),
                      "TS": (c >= 8 and c < 16 and r != 0) or (c >= 16 and not (r in (sh.TA, sh.TQ, sh.TF) or sh.TL0 <= r < sh.TP))}.get(name, False)
            assert broken or unused, (name, r, c)
    # a term's coefficient, a factor's key, where a constraint starts: all KEY material -- another program is another key (below); a value
    # that reaches a factor slot from nowhere has no sender
    i = names.index("EVAL")
    mains[i][0, R.EV_F0 + 8] = (int(mains[i][0, R.EV_F0 + 8]) + 1) % R.P
    assert R.bus_balance(mains, pres, tabs)


def test_air_mode_product_machine_equals_the_restatement_word_for_word():
    from zktls_amd.device import shard_verifier_describe, sha256_air
    import oracle_lib as O
    cases = [(5, 8, 4, 3, 1, O.air_synthetic(8, 3)), (5, 8, 3, 2, 2, R.counter_program(8)), (6, 16, 2, 0, 3, R.counter_program(16)), (7, 64, 5, 1, 1, O.air_synthetic(64, 0)),
             (10, 8, 2, 1, 65, R.counter_program(8)), (6, 640, 2, 1, 1, sha256_air()), (14, 640, 20, 8, 2, sha256_air())]
    for log_n, width, q, pb, nproofs, prog in cases:
        npub = int(prog[4])
        sh = R.Shape(log_n, width, q, pb, npub, nproofs, program=prog)
        names, progs, tabs, pres, h = R.order(sh), R.programs(sh), R.tables(sh), R.preprocessed(sh), R.heights(sh)
        assert len(names) == 9
        for i, nm in enumerate(names):
            p, ln, mw, pw = shard_verifier_describe(log_n, width, q, pb, npub, i, 0, nproofs, program=prog)
            t, _, _, _ = shard_verifier_describe(log_n, width, q, pb, npub, i, 1, nproofs, program=prog)
            e, _, _, _ = shard_verifier_describe(log_n, width, q, pb, npub, i, 2, nproofs, program=prog)
            assert ln == h[nm] and np.array_equal(p, np.asarray(progs[nm], dtype=np.uint32)), (nm, "program", width, nproofs)
            assert np.array_equal(t, np.asarray(tabs[nm], dtype=np.uint32)), (nm, "table", width, nproofs)
            want = np.zeros(0, dtype=np.uint32) if pres[nm] is None else pres[nm].ravel()
            assert np.array_equal(e, want) and pw == (0 if pres[nm] is None else pres[nm].shape[1]), (nm, "preprocessed", width, nproofs)


def test_air_mode_host_key_equals_the_oracles_key(oracle):
    from zktls_amd._lib import Params
    from zktls_amd.device import shard_verifier_key_host
    for log_n, width, q, pb, nproofs, prog in ((5, 8, 4, 3, 1, oracle.air_synthetic(8, 3)), (5, 8, 3, 2, 2, R.counter_program(8)), (6, 16, 2, 0, 3, R.counter_program(16))):
        npub = int(prog[4])
        sh = R.Shape(log_n, width, q, pb, npub, nproofs, program=prog)
        names, pres, h = R.order(sh), R.preprocessed(sh), R.heights(sh)
        pl, lns = [pres[n] for n in names], [h[n] for n in names]
        for blow, nq, pw in ((1, 20, 8), (2, 10, 4)):
            want = [int(x) for x in oracle.machine_setup(pl, lns, oracle.default_params(blow, nq, pw))]
            assert shard_verifier_key_host(log_n, width, q, pb, npub, Params(blow, nq, pw), nproofs, program=prog).tolist() == want, (log_n, width, nproofs, blow)
    # the key is a function of the program: one coefficient changed is another key
    prog = R.counter_program(8)
    other = prog.copy()
    other[-4] = (int(other[-4]) + 1) % R.P
    a = shard_verifier_key_host(5, 8, 3, 2, 3, Params(1, 20, 8), 1, program=prog).tolist()
    assert a != shard_verifier_key_host(5, 8, 3, 2, 3, Params(1, 20, 8), 1, program=other).tolist()
    assert a != shard_verifier_key_host(5, 8, 3, 2, 3, Params(1, 20, 8), 1).tolist()


def test_air_mode_the_oracle_proves_the_join_and_three_verifiers_accept(oracle):
    """two version-7 proofs of one program -> ONE outer proof; checked from (program, public values of both, key): no byte of an inner proof.
    Refused: other public values, the values swapped, another program's key, another key, the version-1 machine's entry"""
    import pyverify_chips
    from zktls_amd._lib import Params
    from zktls_amd.device import verify_shard_recursive, shard_verifier_key_host
    O = oracle
    kind, log_n, width, q, pb = AIR_SHAPES[1]
    oprm, prm = O.default_params(1, 20, 8), Params(1, 20, 8)
    for nproofs in (1, 2):
        made = [air_inner(O, kind, log_n, width, q, pb, shard=p) for p in range(nproofs)]
        prog, proofs, pubs = made[0][0], [m[1] for m in made], [m[2] for m in made]
        sh, mains, pres, progs, tabs, pv = R.machine(proofs, log_n, width, pubs, q, pb, program=prog)
        lns = [m.shape[0].bit_length() - 1 for m in mains]
        widths, pws = [m.shape[1] for m in mains], [0 if p is None else p.shape[1] for p in pres]
        vk = O.machine_setup(pres, lns, oprm)
        op = O.prove_machine_keyed(mains, pres, progs, tabs, pv, oprm)
        assert O.verify_machine_keyed(op, lns, widths, pws, vk, progs, tabs, pv, oprm) == 0
        assert verify_shard_recursive(op, log_n, width, q, pb, pv, vk, prm, n_proofs=nproofs, program=prog) == (0, 0)
        assert pyverify_chips.verify(op.tobytes(), lns, widths, pv, log_blowup=1, num_queries=20, pow_bits=8, programs=progs, tables=tabs, pre_widths=pws, pre_root=[int(x) for x in vk]) is True
        bad = list(pv)
        bad[2] = (bad[2] + 1) % R.P
        assert verify_shard_recursive(op, log_n, width, q, pb, bad, vk, prm, n_proofs=nproofs, program=prog)[0] != 0
        if nproofs == 2:
            assert verify_shard_recursive(op, log_n, width, q, pb, pubs[1] + pubs[0], vk, prm, n_proofs=2, program=prog)[0] != 0
        other = prog.copy()
        other[-4] = (int(other[-4]) + 1) % R.P
        # (a coefficient lives in the KEY alone -- the EVAL chip's preprocessed columns: a verifier takes the key of the program it means, and
        # can derive it without a device)
        assert shard_verifier_key_host(log_n, width, q, pb, len(pubs[0]), prm, nproofs, program=prog).tolist() == [int(x) for x in vk]
        vk_other = shard_verifier_key_host(log_n, width, q, pb, len(pubs[0]), prm, nproofs, program=other)
        assert verify_shard_recursive(op, log_n, width, q, pb, pv, vk_other, prm, n_proofs=nproofs, program=other)[0] != 0
        assert verify_shard_recursive(op, log_n, width, q, pb, pv, (vk + 1) % R.P, prm, n_proofs=nproofs, program=prog)[0] != 0
        assert verify_shard_recursive(op, log_n, width, q, pb, pv, vk, prm, n_proofs=nproofs)[0] != 0


def test_air_mode_a_sha256_proof_in_the_machine(oracle):
    """the SHA-256 chip's program (815 constraints, 5 192 terms, 91 public values, all three selectors) as the inner statement: an oracle-made proof of
    "digest = SHA-256 of a message of 100 bytes" -> the machine's rows hold, every bus balances"""
    import hashlib
    msg = bytes(range(100))
    prog = S.program()
    t, pub = S.trace(S.pad(msg))
    assert S.digest_bytes(pub) == hashlib.sha256(msg).digest()
    log_n, q, pb = t.shape[0].bit_length() - 1, 2, 1
    proof = oracle.prove_shard_air(prog, t, pub, oracle.default_params(1, q, pb)).tobytes()
    sh, mains, pres, progs, tabs, pv = R.machine(proof, log_n, S.WIDTH, pub, q, pb, program=prog)
    assert len(sh.terms) == 5192 and R.heights(sh)["EVAL"] == 13
    for name, main, pre, pr in zip(R.order(sh), mains, pres, progs):
        rows = main if pre is None else np.concatenate([pre, main], axis=1)
        assert S.check_rows(pr, rows, pv) == [], name
    assert R.bus_balance(mains, pres, tabs) == []


def test_the_batch_of_joins_checks_its_arguments_before_it_touches_a_device():
    """zkhip_prove_shard_verifier_batch with no device: a count that is no multiple of the join size, a stride below the size query's answer and a
    null pointer are ZKHIP_ERR_INVALID; a well-formed call is ZKHIP_ERR_NO_DEVICE (there is no CPU fallback)"""
    import ctypes as C
    from zktls_amd import _lib
    from zktls_amd._lib import Params
    L = _lib.load()
    u8p = C.POINTER(C.c_uint8)
    inner, outer = Params(1, 5, 2), Params(1, 20, 8)
    a = np.zeros(64, dtype=np.uint8)
    ptrs = (u8p * 4)(*[a.ctypes.data_as(u8p)] * 4)
    lens = (C.c_size_t * 4)(*[64] * 4)
    pv = np.arange(8, dtype=np.uint32)
    cap = L.zkhip_shard_verifier_proof_size(6, 16, 5, 2, 2, 2, C.byref(outer))
    assert cap > 0
    out = np.zeros(2 * cap, dtype=np.uint8)
    jl = (C.c_size_t * 2)()
    vk = np.zeros(8, dtype=np.uint32)
    u32p = C.POINTER(C.c_uint32)

    def call(n=4, j=2, stride=cap, proofs=ptrs, width=16):
        return L.zkhip_prove_shard_verifier_batch(None, 0, proofs, lens, n, j, 6, width, pv.ctypes.data_as(u32p), 2, C.byref(inner), C.byref(outer), 2, 1,
                                                  out.ctypes.data_as(u8p), stride, jl, vk.ctypes.data_as(u32p))
    assert call(n=3) == -1 and b"multiple" in L.zkhip_last_error()
    assert call(j=0) == -1
    assert call(n=0) == -1
    assert call(stride=cap - 1) == -1 and b"stride" in L.zkhip_last_error()
    assert call(proofs=None) == -1
    assert call(width=12) == -1                                            # (the machine takes widths in multiples of 8: the size query refuses)
    if _lib.device_count() == 0:
        assert call() == -2
