"""The FRI-fold chip (zktls_amd/csrc/fri_chip.hip; SURVEY.md 8f-4, second half: a first step of the recursion behind
SP1ProofMode::Groth16, sp1.rs:116), CPU side: the library's constraint program against the independent Python restatement
(tests/fri_air.py); the view of a golden shard proof as the library's verifier and the pure-Python verifier hand it out; the two-chip
keyed machine on the restated trace under the oracle's prover and three verifiers; and what the machine refuses."""
import hashlib
import json
import os

import numpy as np
import pytest

import fri_air as F
import pyverify
import pyverify_chips
from zktls_amd import _lib
from zktls_amd._lib import Params
from zktls_amd.device import fri_chip_air, fri_view_shard, verify_machine_keyed

P = 2013265921
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "oracle_kat.json")))
GOLDEN = KAT["golden_proof_files"]


def load(name):
    return np.frombuffer(open(os.path.join(HERE, "golden", "proofs", name + ".bin"), "rb").read(), dtype=np.uint8)


def shape_of(traces, pre):
    return ([t.shape[0].bit_length() - 1 for t in traces], [t.shape[1] for t in traces], [0 if p is None else p.shape[1] for p in pre])


@pytest.mark.parametrize("layers", [2, 3, 6, 10, 14, 20, 22])
def test_program_words_equal_the_python_restatement(oracle, layers):
    prog = fri_chip_air(layers)
    mine = F.program(layers)
    assert prog.tolist() == mine.tolist()
    assert oracle.air_validate(prog, F.width_of(layers), F.n_public_of(layers)) == 1
    assert oracle.air_log_quotient_degree(prog) == 1                 # degree 3 with the selectors: two quotient chunks, blowup 2 is enough
    lib = _lib.load()
    assert lib.zkhip_fri_chip_width(layers) == F.width_of(layers) and lib.zkhip_fri_chip_air(1, None, 0) == 0 and lib.zkhip_fri_chip_air(23, None, 0) == 0


@pytest.mark.parametrize("name", ["v1_6x8", "v1_10x16"])
def test_view_of_a_golden_proof_agrees_between_the_library_and_the_python_verifier(name):
    g = GOLDEN[name]
    b = load(name)
    view = {}
    assert pyverify.verify(b.tobytes(), g["log_n"], g["width"], g["public"], *g["shape"], view=view) is True
    lib_view = fri_view_shard(b, g["log_n"], g["width"], g["public"], Params(*g["shape"]))
    assert lib_view["betas"] == view["betas"] and lib_view["final"] == view["final"]
    assert [(q[0], list(q[1]), [list(s) for s in q[2]]) for q in view["queries"]] == lib_view["queries"]
    assert len(view["queries"]) == g["shape"][1] and len(view["betas"]) == g["log_n"]
    # a rejected proof has no view
    bad = b.copy()
    bad[-5] ^= 1
    with pytest.raises(_lib.ZkHipError):
        fri_view_shard(bad, g["log_n"], g["width"], g["public"], Params(*g["shape"]))
    # the chains of the view fold to its final value (what the chip will prove), by the restatement's own arithmetic
    tr, final = F.trace(view)
    assert list(final) == view["final"]


@pytest.mark.parametrize("name", ["v1_6x8", "v1_10x16"])
def test_the_challenges_of_a_golden_proof_are_a_sponge_chain_over_its_layer_roots(name):
    """zkhip_fri_view_transcript: roots, challenges and the duplex challenger's capacity as the commit phase finds it.  By the
    independent Poseidon2 of tests/pyref.py every challenge is ONE sponge step: state <- (root_l | capacity), permute,
    beta_l = (state[7], state[6], state[5], state[4]) -- the statement a transcript chip will prove (docs/RECURSION_NEXT.md)"""
    import pyref
    from zktls_amd.device import fri_view_transcript
    g = GOLDEN[name]
    b = load(name)
    prm = Params(*g["shape"])
    roots, betas, capacity, pending = fri_view_transcript(b, g["log_n"], g["width"], g["public"], prm)
    view = fri_view_shard(b, g["log_n"], g["width"], g["public"], prm)
    assert betas == view["betas"] and pending == 0 and len(roots) == g["log_n"]
    cap = list(capacity)
    for root, beta in zip(roots, betas):
        state = pyref.poseidon2(list(root) + cap)
        assert [state[7], state[6], state[5], state[4]] == list(beta)
        cap = state[8:]
    bad = b.copy()
    bad[-9] ^= 4
    with pytest.raises(_lib.ZkHipError):
        fri_view_transcript(bad, g["log_n"], g["width"], g["public"], prm)


@pytest.mark.parametrize("name,shape", [("v1_6x8", (1, 12, 4)), ("v1_10x16", (1, 10, 6)), ("v1_10x16", (2, 7, 0))])
def test_machine_of_a_golden_proofs_view_under_the_oracle_prover_and_three_verifiers(oracle, name, shape):
    O = oracle
    g = GOLDEN[name]
    view = fri_view_shard(load(name), g["log_n"], g["width"], g["public"], Params(*g["shape"]))
    traces, pre, progs, tables, pub = F.machine(view)
    lns, ws, pws = shape_of(traces, pre)
    oprm, prm = O.default_params(*shape), Params(*shape)
    root = O.machine_setup(pre, lns, oprm)
    proof = O.prove_machine_keyed(traces, pre, progs, tables, pub, oprm)
    assert O.verify_machine_keyed(proof, lns, ws, pws, root, progs, tables, pub, oprm) == 0
    assert verify_machine_keyed(proof, lns, ws, pws, root, progs, tables, pub, prm) == (0, 0)
    assert pyverify_chips.verify(proof.tobytes(), lns, ws, pub, shape[0], shape[1], shape[2], programs=progs, tables=tables, pre_widths=pws, pre_root=[int(v) for v in root]) is True


def _machine_rejected(O, traces, pre, progs, tables, pub, shape, root=None):
    """the oracle proves whatever it is given; an unsatisfied constraint or an unbalanced bus shows in the verifiers"""
    lns, ws, pws = shape_of(traces, pre)
    oprm = O.default_params(*shape)
    root = O.machine_setup(pre, lns, oprm) if root is None else root
    proof = O.prove_machine_keyed(traces, pre, progs, tables, pub, oprm)
    a = O.verify_machine_keyed(proof, lns, ws, pws, root, progs, tables, pub, oprm)
    b = verify_machine_keyed(proof, lns, ws, pws, root, progs, tables, pub, Params(*shape))
    assert (a != 0) == (b[0] != 0)
    return a != 0


def test_what_the_machine_refuses(oracle):
    O = oracle
    shape = (1, 16, 2)
    view = F.random_view(5, 9, seed=4)
    traces, pre, progs, tables, pub = F.machine(view)
    assert not _machine_rejected(O, traces, pre, progs, tables, pub, shape)
    honest_root = O.machine_setup(pre, shape_of(traces, pre)[0], O.default_params(*shape))
    R = 5

    def tampered(fn):
        t = [x.copy() for x in traces]
        fn(t[0])
        return t
    # a wrong fold; the other square root for X (with its inverse and square kept consistent); a sibling that is not the listed one
    assert _machine_rejected(O, tampered(lambda t: t.__setitem__((3, F.FOLD), (int(t[3, F.FOLD]) + 1) % P)), pre, progs, tables, pub, shape)

    def flip_x(t):
        t[0, F.X] = (P - int(t[0, F.X])) % P
        t[0, F.XI] = (P - int(t[0, F.XI])) % P
    assert _machine_rejected(O, tampered(flip_x), pre, progs, tables, pub, shape)
    assert _machine_rejected(O, tampered(lambda t: t.__setitem__((2 * R + 1, F.E1 + 2), (int(t[2 * R + 1, F.E1 + 2]) + 1) % P)), pre, progs, tables, pub, shape)
    # a query left out: its rows turned into padding -- the table's multiplicities are part of the KEY, so the buses no longer balance
    def drop_last_query(t):
        rows = slice(8 * R, 9 * R)
        t[rows] = 0
        t[rows, F.T] = 1
    assert _machine_rejected(O, tampered(drop_last_query), pre, progs, tables, pub, shape)
    # ... and a prover who ALSO rewrites the table (fewer reads listed) is proving against another key
    fewer = {"betas": view["betas"], "queries": view["queries"][:8]}
    t2, pre2, _, _, pub2 = F.machine(fewer)
    if t2[0].shape == traces[0].shape:
        assert _machine_rejected(O, t2, pre2, progs, tables, pub2, shape, root=honest_root)
    # a chain that ends elsewhere: the public final value is what every END row must show
    pub_bad = list(pub)
    pub_bad[-1] = (pub_bad[-1] + 1) % P
    assert _machine_rejected(O, traces, pre, progs, tables, pub_bad, shape)


def test_entry_point_argument_checks():
    lib = _lib.load()
    prm = Params(1, 8, 2)
    assert lib.zkhip_fri_indices_proof_size(1, 8, 0, prm) == 0 and lib.zkhip_fri_indices_proof_size(6, 0, 0, prm) == 0
    assert lib.zkhip_fri_indices_proof_size(6, 8, 0, prm) > 0
    g = GOLDEN["v3_r0_9x8"]                       # fold by 16: no view
    with pytest.raises(_lib.ZkHipError):
        fri_view_shard(load("v3_r0_9x8"), g["log_n"], g["width"], g["public"], Params(*g["shape"]))


# ------------------------------------------------------------------ the wired machine: Merkle paths of the pairs in-circuit
@pytest.mark.parametrize("layers", [2, 6, 10, 20])
def test_wired_programs_equal_the_python_restatements(oracle, layers):
    import poseidon2_air as P2
    from zktls_amd.device import fri_layers_programs
    p2f, fri = fri_layers_programs(layers)
    assert p2f.tolist() == P2.program(fri_layers=True, n_public=F.n_public_of(layers)).tolist()
    assert fri.tolist() == F.program(layers, wired=True).tolist()
    for prog, w in ((p2f, 360), (fri, F.width_of(layers, True))):
        assert oracle.air_validate(prog, w, F.n_public_of(layers)) == 1 and oracle.air_log_quotient_degree(prog) == 1


@pytest.mark.parametrize("name,shape", [("v1_6x8", (1, 10, 4)), ("v1_10x16", (1, 8, 6))])
def test_wired_machine_of_a_golden_proofs_view(oracle, name, shape):
    """view with roots and paths (library == Python verifier), the four-chip machine on the restated arrays under the oracle's prover and
    three verifiers, and what it refuses: a sibling digest that is not the committed one, a pair that is not the opened one, a query
    left out, other roots in the key"""
    from zktls_amd.device import fri_view_shard_paths
    O = oracle
    g = GOLDEN[name]
    b = load(name)
    view = fri_view_shard_paths(b, g["log_n"], g["width"], g["public"], Params(*g["shape"]))
    pview = {}
    assert pyverify.verify(b.tobytes(), g["log_n"], g["width"], g["public"], *g["shape"], view=pview) is True
    assert pview["roots"] == view["roots"] and pview["paths"] == view["paths"] and pview["betas"] == view["betas"]
    traces, pre, progs, tables, pub = F.machine_layers(view)
    lns, ws, pws = shape_of(traces, pre)
    assert ws[0] == 360 and pws == [0, 0, 8, 12] and lns == sorted(lns, reverse=True)
    oprm, prm = O.default_params(*shape), Params(*shape)
    root = O.machine_setup(pre, lns, oprm)
    proof = O.prove_machine_keyed(traces, pre, progs, tables, pub, oprm)
    nq = len(view["queries"])
    assert O.verify_machine_keyed(proof, lns, ws, pws, root, progs, tables, pub, oprm) == 0
    assert verify_machine_keyed(proof, lns, ws, pws, root, progs, tables, pub, prm) == (0, 0)
    assert pyverify_chips.verify(proof.tobytes(), lns, ws, pub, shape[0], shape[1], shape[2], programs=progs, tables=tables, pre_widths=pws,
                                 pre_root=[int(v) for v in root]) is True
    import poseidon2_air as P2
    R = g["log_n"]

    def tampered(chip, fn):
        t = [x.copy() for x in traces]
        fn(t[chip])
        return t
    # a sibling digest on a path that is not the committed one: the END row's digest is no longer the layer's root -> the ROOTS bus does not balance
    assert _machine_rejected(O, tampered(0, lambda t: t.__setitem__((1, P2.IN + 8), (int(t[1, P2.IN + 8]) + 1) % P)), pre, progs, tables, pub, shape)
    # the fold chip claims another pair than the one whose path is shown
    assert _machine_rejected(O, tampered(1, lambda t: t.__setitem__((R + 2, F.E0 + 1), (int(t[R + 2, F.E0 + 1]) + 1) % P)), pre, progs, tables, pub, shape)
    # a path shown for another leaf index than the one the fold chip uses
    assert _machine_rejected(O, tampered(0, lambda t: t.__setitem__((0, P2.KP), (int(t[0, P2.KP]) + 2) % P)), pre, progs, tables, pub, shape)

    # a query left out of the fold chip: the QUERIES table still expects its start
    def drop(t):
        t[(nq - 1) * R:nq * R] = 0
        t[(nq - 1) * R:nq * R, F.T] = 1
    assert _machine_rejected(O, tampered(1, drop), pre, progs, tables, pub, shape)
    # other roots in the key
    pre2 = [None, None, pre[2], pre[3].copy()]
    pre2[3][0, 1] = (int(pre2[3][0, 1]) + 1) % P
    assert _machine_rejected(O, traces, pre2, progs, tables, pub, shape)


# ------------------------------------------------------------------ the transcript machine: the challenges derived in-circuit
@pytest.mark.parametrize("name,shape", [("v1_6x8", (1, 10, 4)), ("v1_10x16", (1, 8, 6))])
def test_transcript_machine_of_a_golden_proofs_view(oracle, name, shape):
    """the wired machine whose Poseidon2 chip starts with transcript rows: a sponge chain over the layer roots, from the challenger's
    capacity (public), whose outputs reach the fold rows through the ROOTS table's main columns: the challenges are neither public
    values nor part of the key.  Under the oracle's prover and three verifiers; refused: a challenge the chain does not produce, fold
    rows with another challenge, another capacity, other roots, a root absorbed that is not the layer's, a transcript row left out"""
    from zktls_amd.device import fri_view_shard_paths, fri_view_transcript
    import poseidon2_air as P2
    O = oracle
    g = GOLDEN[name]
    b = load(name)
    prm0 = Params(*g["shape"])
    view = fri_view_shard_paths(b, g["log_n"], g["width"], g["public"], prm0)
    roots, betas, capacity, pending = fri_view_transcript(b, g["log_n"], g["width"], g["public"], prm0)
    assert roots == view["roots"] and betas == view["betas"] and pending == 0
    traces, pre, progs, tables, pub = F.machine_layers(view, capacity=capacity)
    lns, ws, pws = shape_of(traces, pre)
    R = g["log_n"]
    assert ws[0] == P2.WIDTH_T and ws[3] == 8 and pws == [0, 0, 8, 12] and lns == sorted(lns, reverse=True) and len(pub) == 12
    assert all(int(traces[0][l, P2.TRS]) == 1 and int(traces[0][l, P2.LNP]) == l for l in range(R)) and int(traces[0][R, P2.TRS]) == 0
    for prog, w in zip(progs, ws):
        assert O.air_log_quotient_degree(prog) == 1
    oprm, prm = O.default_params(*shape), Params(*shape)
    root = O.machine_setup(pre, lns, oprm)
    proof = O.prove_machine_keyed(traces, pre, progs, tables, pub, oprm)
    assert O.verify_machine_keyed(proof, lns, ws, pws, root, progs, tables, pub, oprm) == 0
    assert verify_machine_keyed(proof, lns, ws, pws, root, progs, tables, pub, prm) == (0, 0)
    assert pyverify_chips.verify(proof.tobytes(), lns, ws, pub, shape[0], shape[1], shape[2], programs=progs, tables=tables, pre_widths=pws,
                                 pre_root=[int(v) for v in root]) is True
    # the library builds the same two chip programs
    from zktls_amd.device import fri_transcript_programs
    p2t, frit = fri_transcript_programs(R)
    assert p2t.tolist() == progs[0].tolist() and frit.tolist() == progs[1].tolist()
    nq = len(view["queries"])

    def tampered(chip, fn):
        t = [x.copy() for x in traces]
        fn(t[chip])
        return t
    # a challenge in the ROOTS table's main columns that the sponge chain does not produce (the fold rows would take it from there)
    assert _machine_rejected(O, tampered(3, lambda t: t.__setitem__((1, 2), (int(t[1, 2]) + 1) % P)), pre, progs, tables, pub, shape)
    # fold rows that use another challenge than the table's
    assert _machine_rejected(O, tampered(1, lambda t: t.__setitem__((R + 1, F.BETA), (int(t[R + 1, F.BETA]) + 1) % P)), pre, progs, tables, pub, shape)
    # another capacity than the public one on the first transcript row
    pub2 = list(pub)
    pub2[-3] = (pub2[-3] + 1) % P
    assert _machine_rejected(O, traces, pre, progs, tables, pub2, shape)
    # other roots in the key: the transcript rows absorb what the table does not hold
    pre2 = [None, None, pre[2], pre[3].copy()]
    pre2[3][1, 4] = (int(pre2[3][1, 4]) + 1) % P
    assert _machine_rejected(O, traces, pre2, progs, tables, pub, shape)
    # a transcript row that absorbs something else than its layer's root
    assert _machine_rejected(O, tampered(0, lambda t: t.__setitem__((2, P2.IN + 3), (int(t[2, P2.IN + 3]) + 1) % P)), pre, progs, tables, pub, shape)
    # a transcript row that claims another layer number
    assert _machine_rejected(O, tampered(0, lambda t: t.__setitem__((1, P2.LNP), 3)), pre, progs, tables, pub, shape)
    # the last transcript row left out (its flag cleared): the ROOTS table still expects its root and its challenge
    assert _machine_rejected(O, tampered(0, lambda t: t.__setitem__((R - 1, P2.TRS), 0)), pre, progs, tables, pub, shape)


# ------------------------------------------------------------------ the query-phase machine: proof of work and query indices in-circuit
@pytest.mark.parametrize("name,shape", [("v1_6x8", (1, 10, 4)), ("v1_10x16", (1, 8, 6))])
def test_query_phase_machine_of_a_golden_proofs_view(oracle, name, shape):
    """the transcript machine whose sponge chain goes on through the final value and the proof-of-work witness: the words it then hands
    out are decomposed by a fifth chip (SAMPLES) -- the first one's low bits must be zero (proof of work), the others' low bits are the
    query indices, which reach the QUERIES table's MAIN column and from there the fold rows.  The key holds no index any more.  Under
    the oracle's prover and three verifiers; refused: another index in the table, a query walked from another index, another witness,
    a word that is not the sponge's, a non-canonical decomposition, a proof-of-work word with a low bit set, a query row left out"""
    from zktls_amd.device import fri_view_shard_paths, fri_view_transcript, fri_view_witness
    import poseidon2_air as P2
    O = oracle
    g = GOLDEN[name]
    b = load(name)
    prm0 = Params(*g["shape"])
    view = fri_view_shard_paths(b, g["log_n"], g["width"], g["public"], prm0)
    _, _, capacity, _ = fri_view_transcript(b, g["log_n"], g["width"], g["public"], prm0)
    witness = fri_view_witness(b, g["log_n"], g["width"], g["public"], prm0)
    pow_bits, nq, R = g["shape"][2], len(view["queries"]), g["log_n"]
    traces, pre, progs, tables, pub = F.machine_layers(view, capacity=capacity, query_phase=(witness, pow_bits))
    lns, ws, pws = shape_of(traces, pre)
    assert ws == [P2.WIDTH_T, F.width_of(R, True), 4, 8, F.S_MAIN] and pws == [0, 0, 8, 12, F.S_PRE] and lns == sorted(lns, reverse=True) and len(pub) == 12
    n_rows = F.sample_rows(nq)
    assert all(int(traces[0][R + i, P2.QP]) == 1 and int(traces[0][R + i, P2.LNP]) == R + i for i in range(n_rows)) and int(traces[0][R + n_rows, P2.QP]) == 0
    assert [int(v) for v in pre[2][:nq, 0]] == list(range(nq))                     # the key lists the queries by number, not by index
    for prog in progs:
        assert O.air_log_quotient_degree(prog) == 1
    oprm, prm = O.default_params(*shape), Params(*shape)
    root = O.machine_setup(pre, lns, oprm)
    proof = O.prove_machine_keyed(traces, pre, progs, tables, pub, oprm)
    assert O.verify_machine_keyed(proof, lns, ws, pws, root, progs, tables, pub, oprm) == 0
    assert verify_machine_keyed(proof, lns, ws, pws, root, progs, tables, pub, prm) == (0, 0)
    assert pyverify_chips.verify(proof.tobytes(), lns, ws, pub, shape[0], shape[1], shape[2], programs=progs, tables=tables, pre_widths=pws,
                                 pre_root=[int(v) for v in root]) is True

    # the library builds the same programs, and its entry point for this machine accepts the oracle's proof
    from zktls_amd.device import fri_indices_programs, fri_transcript_programs, verify_fri_indices
    p2q, smp = fri_indices_programs(R, pow_bits)
    assert p2q.tolist() == progs[0].tolist() and smp.tolist() == progs[4].tolist() and fri_transcript_programs(R)[1].tolist() == progs[1].tolist()
    assert verify_fri_indices(proof, view["final"], capacity, R, nq, pow_bits, root, prm) == (0, 0)          # neither a challenge nor an index is handed over
    assert verify_fri_indices(proof, view["final"], capacity, R, nq, pow_bits + 1, root, prm)[0] == -6       # (another proof-of-work claim: another program)
    other = list(view["final"])
    other[2] = (other[2] + 1) % P
    assert verify_fri_indices(proof, other, capacity, R, nq, pow_bits, root, prm)[0] == -6

    def tampered(chip, fn):
        t = [x.copy() for x in traces]
        fn(t[chip])
        return t

    def bump(r, c, d=1):
        return lambda t: t.__setitem__((r, c), (int(t[r, c]) + d) % P)
    # another index in the QUERIES table than the one the SAMPLES chip derives
    assert _machine_rejected(O, tampered(2, bump(1, 0)), pre, progs, tables, pub, shape)
    # a query whose fold rows start from another index
    assert _machine_rejected(O, tampered(1, bump(R, F.IDX)), pre, progs, tables, pub, shape)
    # another witness on the first query row
    assert _machine_rejected(O, tampered(0, bump(R, P2.IN + 4)), pre, progs, tables, pub, shape)
    # the first query row does not keep the chain's rate word 5 / the second does not start from the first's output
    assert _machine_rejected(O, tampered(0, bump(R, P2.IN + 5)), pre, progs, tables, pub, shape)
    assert _machine_rejected(O, tampered(0, bump(R + 1, P2.IN + 2)), pre, progs, tables, pub, shape)
    # a sampled word in the SAMPLES chip that is not the sponge's
    assert _machine_rejected(O, tampered(4, bump(0, F.S_W + 3)), pre, progs, tables, pub, shape)

    # a decomposition of w + P instead of w (non-canonical): find a word below 2^31 - P
    def noncanonical(t):
        for j in range(1, 8):
            w = int(t[0, F.S_W + j])
            if w + P < (1 << 31):
                v = w + P
                bits = [(v >> i) & 1 for i in range(31)]
                t[0, F.S_BITS + 31 * j:F.S_BITS + 31 * j + 31] = bits
                t[0, F.S_H1 + j], t[0, F.S_H2 + j], t[0, F.S_HH + j] = bits[30] & bits[29], bits[28] & bits[27], bits[30] & bits[29] & bits[28] & bits[27]
                t[0, F.S_IDX + j] = v & ((1 << (R + 1)) - 1)
                return
        raise AssertionError("no small word on the first row")
    try:
        bad = tampered(4, noncanonical)
    except AssertionError:
        bad = None                                                                  # (w < 2^27 - 1 happens with probability 1/15 per word)
    if bad is not None:
        assert _machine_rejected(O, bad, pre, progs, tables, pub, shape)
    # a proof-of-work word with a low bit set (and the word adjusted to match its bits): the sponge's word is another
    if pow_bits:
        def pow_bit(t):
            t[0, F.S_BITS] = 1
            t[0, F.S_W] = (int(t[0, F.S_W]) + 1) % P
        assert _machine_rejected(O, tampered(4, pow_bit), pre, progs, tables, pub, shape)
    # the last query row left out: the SAMPLES chip still expects its words
    assert _machine_rejected(O, tampered(0, lambda t: t.__setitem__((R + n_rows - 1, P2.QP), 0)), pre, progs, tables, pub, shape)
    # another final value than the one absorbed
    pub2 = list(pub)
    pub2[1] = (pub2[1] + 1) % P
    assert _machine_rejected(O, traces, pre, progs, tables, pub2, shape)


@pytest.mark.parametrize("layers,pow_bits", [(2, 0), (6, 4), (20, 16), (22, 30)])
def test_query_phase_programs_equal_the_python_restatements(oracle, layers, pow_bits):
    import poseidon2_air as P2
    from zktls_amd.device import fri_indices_programs
    p2q, smp = fri_indices_programs(layers, pow_bits)
    assert p2q.tolist() == P2.program(fri_layers=True, n_public=F.N_PUBLIC_T, transcript=4, queries=0).tolist()
    assert smp.tolist() == F.samples_program(layers, 100, pow_bits, F.N_PUBLIC_T).tolist()
    for prog, w in ((p2q, P2.WIDTH_T), (smp, F.S_PRE + F.S_MAIN)):
        assert oracle.air_validate(prog, w, F.N_PUBLIC_T) == 1 and oracle.air_log_quotient_degree(prog) == 1
    lib = _lib.load()
    prm = Params(1, 8, 2)
    assert lib.zkhip_fri_indices_proof_size(layers, 100, pow_bits, prm) > 0
    assert lib.zkhip_fri_indices_proof_size(layers, 100, 31, prm) == 0 and lib.zkhip_fri_indices_proof_size(layers, 100, -1, prm) == 0
    assert lib.zkhip_fri_indices_program(2, layers, pow_bits, None, 0) == 0 and lib.zkhip_fri_indices_program(0, 1, pow_bits, None, 0) == 0


@pytest.mark.parametrize("name", ["v1_6x8", "v1_10x16"])
def test_the_one_pass_view_equals_the_three_views(name):
    from zktls_amd.device import fri_view_all, fri_view_shard_paths, fri_view_transcript, fri_view_witness
    g = GOLDEN[name]
    b = load(name)
    prm0 = Params(*g["shape"])
    view, cap, wit = fri_view_all(b, g["log_n"], g["width"], g["public"], prm0)
    assert view == fri_view_shard_paths(b, g["log_n"], g["width"], g["public"], prm0)
    assert cap == fri_view_transcript(b, g["log_n"], g["width"], g["public"], prm0)[2] and wit == fri_view_witness(b, g["log_n"], g["width"], g["public"], prm0)
    bad = b.copy()
    bad[len(bad) // 3] ^= 4
    with pytest.raises(_lib.ZkHipError):
        fri_view_all(bad, g["log_n"], g["width"], g["public"], prm0)
