"""The FRI-fold chip on the GPU (zkhip_prove_fri_queries; SURVEY.md 8f-4, second half): a shard is proven on the device, the verifier
hands out the view of its FRI part, and a second proof -- a keyed machine of the FRI-fold chip and the preprocessed OPENINGS table --
states that every query chain of that view folds to the final value.  The chip's trace comes from the device generator and must equal
the independent Python restatement (tests/fri_air.py) word for word; the machine proof must equal, byte for byte, what the oracle's
generic keyed-machine prover makes of the restated arrays; three verifiers accept it."""
import time

import numpy as np
import pytest

import fri_air as F
import pyverify
import pyverify_chips
from zktls_amd._lib import Params
from zktls_amd.device import fri_view_shard, verify_fri_queries, verify_machine_keyed, verify_shard

pytestmark = pytest.mark.gpu
SEED = 0x5A4B544C53
P = 2013265921


def shape_of(traces, pre):
    return ([t.shape[0].bit_length() - 1 for t in traces], [t.shape[1] for t in traces], [0 if p is None else p.shape[1] for p in pre])


@pytest.mark.parametrize("log_n,width,inner,outer", [(6, 8, (1, 9, 4), (1, 12, 4)), (10, 16, (1, 25, 8), (1, 20, 8)), (13, 32, (1, 40, 8), (2, 12, 0))])
def test_fri_queries_of_a_shard_proof_fold_in_circuit(ctx, oracle, log_n, width, inner, outer):
    O = oracle
    iprm, oprm, prm = Params(*inner), O.default_params(*outer), Params(*outer)
    pv = [7, 8, 9]
    trace = ctx.gen_trace(SEED, 21, log_n, width)
    shard_proof = ctx.prove_shard(trace, log_n, width, pv, iprm)
    assert verify_shard(shard_proof, log_n, width, pv, iprm) == (0, 0)
    view = fri_view_shard(shard_proof, log_n, width, pv, iprm)
    pview = {}
    assert pyverify.verify(shard_proof.tobytes(), log_n, width, pv, *inner, view=pview) is True
    assert pview["betas"] == view["betas"] and pview["final"] == view["final"]
    assert [(q[0], list(q[1]), [list(s) for s in q[2]]) for q in pview["queries"]] == view["queries"]
    # the chip's trace: device generator == Python restatement
    traces, pre, progs, tables, pub = F.machine(view)
    lns, ws, pws = shape_of(traces, pre)
    d_trace, finals = ctx.fri_chip_gen_trace(view, lns[0])
    assert (d_trace.download().reshape(-1, ws[0]) == traces[0]).all()
    assert (finals == np.array(view["final"], dtype=np.uint32)).all()
    d_trace.free()
    # key and proof: bytes equal to the oracle's generic keyed-machine prover on the restated arrays
    key, final = ctx.fri_queries_key(view, prm)
    assert final == view["final"]
    assert key.root.tolist() == O.machine_setup(pre, lns, oprm).tolist()
    proof = ctx.prove_fri_queries(key, view, prm)
    oproof = O.prove_machine_keyed(traces, pre, progs, tables, pub, oprm)
    assert proof.tobytes() == oproof.tobytes(), "FRI-queries machine proof differs from the oracle's"
    # three verifiers
    nq = len(view["queries"])
    assert verify_fri_queries(proof, view["betas"], view["final"], nq, key.root, prm) == (0, 0)
    assert verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, prm) == (0, 0)
    assert O.verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, oprm) == 0
    assert pyverify_chips.verify(proof.tobytes(), lns, ws, pub, outer[0], outer[1], outer[2], programs=progs, tables=tables, pre_widths=pws,
                                 pre_root=[int(v) for v in key.root]) is True
    # the statement is about THIS proof's view: another proof of the same shape has another key and other challenges
    other = ctx.prove_shard(trace, log_n, width, [7, 8, 10], iprm)
    oview = fri_view_shard(other, log_n, width, [7, 8, 10], iprm)
    okey, _ = ctx.fri_queries_key(oview, prm)
    assert okey.root.tolist() != key.root.tolist()
    assert verify_fri_queries(proof, oview["betas"], oview["final"], nq, okey.root, prm)[0] == -6
    assert verify_fri_queries(proof, view["betas"], view["final"], nq, okey.root, prm)[0] == -6
    # a view that was tampered with (a sibling changed) is refused before anything is proven: its chains no longer end in one value
    bad = {"betas": view["betas"], "queries": [(q[0], q[1], [list(s) for s in q[2]]) for q in view["queries"]]}
    bad["queries"][0][2][1][0] = (bad["queries"][0][2][1][0] + 1) % P
    with pytest.raises(Exception):
        ctx.prove_fri_queries(key, bad, prm)
    key.close()
    okey.close()
    trace.free()


def test_fullsize_fri_queries_of_the_headline_shard(ctx, oracle):
    """the 100 queries x 20 layers of a 2^20 x 256 SP1-shape shard proof: 2 000 rows of the chip, a 2^11-row machine"""
    O = oracle
    log_n, width = 20, 256
    iprm, prm, oprm = Params(1, 100, 16), Params(1, 100, 16), O.default_params(1, 100, 16)
    trace = ctx.gen_trace(SEED, 31, log_n, width)
    shard_proof = ctx.prove_shard(trace, log_n, width, [1, 2, 3], iprm)
    trace.free()
    t0 = time.perf_counter()
    view = fri_view_shard(shard_proof, log_n, width, [1, 2, 3], iprm)
    t1 = time.perf_counter()
    key, final = ctx.fri_queries_key(view, prm)
    t2 = time.perf_counter()
    proof = ctx.prove_fri_queries(key, view, prm)
    t3 = time.perf_counter()
    proof = ctx.prove_fri_queries(key, view, prm)
    t4 = time.perf_counter()
    assert verify_fri_queries(proof, view["betas"], view["final"], 100, key.root, prm) == (0, 0)
    t5 = time.perf_counter()
    print("\nFRI queries of a 2^20 x 256 shard proof: view (host verifier) %.1f ms, key %.1f ms, machine proof %.1f ms (first %.1f), %d bytes, host verification %.1f ms"
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t4 - t3) * 1e3, (t3 - t2) * 1e3, proof.size, (t5 - t4) * 1e3))
    traces, pre, progs, tables, pub = F.machine(view)
    lns, ws, pws = shape_of(traces, pre)
    assert lns == [11, 11] and ws[0] == 52
    assert proof.tobytes() == O.prove_machine_keyed(traces, pre, progs, tables, pub, oprm).tobytes()
    assert O.verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, oprm) == 0
    key.close()


# ------------------------------------------------------------------ the wired machine: Merkle paths of the pairs in-circuit
@pytest.mark.parametrize("log_n,width,inner,outer", [(6, 8, (1, 9, 4), (1, 12, 4)), (11, 16, (1, 20, 8), (1, 16, 6))])
def test_fri_layers_of_a_shard_proof_open_and_fold_in_circuit(ctx, oracle, log_n, width, inner, outer):
    """zkhip_prove_fri_layers: the Poseidon2 chip's FRI-layers variant authenticates every layer pair of every query against the layer's
    root, the fold chip folds them; both traces come from the device generators and equal the Python restatements; the four-chip machine
    proof equals the oracle's bytes; three verifiers accept it; the key holds only (index, reduced opening) and the layer roots."""
    from zktls_amd.device import fri_view_shard_paths, verify_fri_layers
    O = oracle
    iprm, oprm, prm = Params(*inner), O.default_params(*outer), Params(*outer)
    pv = [3, 1, 4]
    trace = ctx.gen_trace(SEED, 41, log_n, width)
    shard_proof = ctx.prove_shard(trace, log_n, width, pv, iprm)
    trace.free()
    view = fri_view_shard_paths(shard_proof, log_n, width, pv, iprm)
    pview = {}
    assert pyverify.verify(shard_proof.tobytes(), log_n, width, pv, *inner, view=pview) is True
    assert pview["roots"] == view["roots"] and pview["paths"] == view["paths"]
    traces, pre, progs, tables, pub = F.machine_layers(view)
    lns, ws, pws = shape_of(traces, pre)
    d_p2 = ctx.fri_layers_gen_paths_trace(view, lns[0])
    assert (d_p2.download().reshape(-1, 360) == traces[0]).all()
    d_p2.free()
    key = ctx.fri_layers_key(view, prm)
    assert key.root.tolist() == O.machine_setup(pre, lns, oprm).tolist()
    proof = ctx.prove_fri_layers(key, view, prm)
    oproof = O.prove_machine_keyed(traces, pre, progs, tables, pub, oprm)
    assert proof.tobytes() == oproof.tobytes(), "FRI-layers machine proof differs from the oracle's"
    nq = len(view["queries"])
    assert verify_fri_layers(proof, view["betas"], view["final"], nq, key.root, prm) == (0, 0)
    assert verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, prm) == (0, 0)
    assert O.verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, oprm) == 0
    assert pyverify_chips.verify(proof.tobytes(), lns, ws, pub, outer[0], outer[1], outer[2], programs=progs, tables=tables, pre_widths=pws,
                                 pre_root=[int(v) for v in key.root]) is True
    # a path that does not end in its layer's root is refused before anything is proven
    bad = dict(view)
    bad["paths"] = [[[list(d) for d in layer] for layer in pq] for pq in view["paths"]]
    bad["paths"][1][0][0][3] = (bad["paths"][1][0][0][3] + 1) % P
    with pytest.raises(Exception):
        ctx.prove_fri_layers(key, bad, prm)
    # a key made from other reduced openings does not accept this proof
    other = dict(view)
    other["queries"] = [(q[0], [(q[1][0] + 1) % P] + list(q[1][1:]), q[2]) for q in view["queries"]]
    okey = ctx.fri_layers_key(other, prm)
    assert okey.root.tolist() != key.root.tolist()
    assert verify_fri_layers(proof, view["betas"], view["final"], nq, okey.root, prm)[0] == -6
    key.close()
    okey.close()


def test_fullsize_fri_layers_of_the_headline_shard(ctx, oracle):
    """100 queries x 20 layers of a 2^20 x 256 shard proof: 23 000 rows of Poseidon2 permutations (2^15 x 360), 2 000 rows of folds"""
    from zktls_amd.device import fri_view_shard_paths, verify_fri_layers
    O = oracle
    log_n, width = 20, 256
    iprm, prm, oprm = Params(1, 100, 16), Params(1, 100, 16), O.default_params(1, 100, 16)
    trace = ctx.gen_trace(SEED, 32, log_n, width)
    shard_proof = ctx.prove_shard(trace, log_n, width, [1, 2, 3], iprm)
    trace.free()
    t0 = time.perf_counter()
    view = fri_view_shard_paths(shard_proof, log_n, width, [1, 2, 3], iprm)
    t1 = time.perf_counter()
    key = ctx.fri_layers_key(view, prm)
    t2 = time.perf_counter()
    proof = ctx.prove_fri_layers(key, view, prm)
    t3 = time.perf_counter()
    proof = ctx.prove_fri_layers(key, view, prm)
    t4 = time.perf_counter()
    assert verify_fri_layers(proof, view["betas"], view["final"], 100, key.root, prm) == (0, 0)
    t5 = time.perf_counter()
    print("\nFRI layers of a 2^20 x 256 shard proof (Merkle paths + folds in-circuit): view %.1f ms, key %.1f ms, machine proof %.1f ms (first %.1f), %d bytes, host verification %.1f ms"
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t4 - t3) * 1e3, (t3 - t2) * 1e3, proof.size, (t5 - t4) * 1e3))
    traces, pre, progs, tables, pub = F.machine_layers(view)
    lns, ws, pws = shape_of(traces, pre)
    assert lns == [15, 11, 7, 5]
    assert proof.tobytes() == O.prove_machine_keyed(traces, pre, progs, tables, pub, oprm).tobytes()
    assert O.verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, oprm) == 0
    key.close()


# ------------------------------------------------------------------ the transcript machine: the challenges derived in-circuit
@pytest.mark.parametrize("log_n,width,inner,outer", [(6, 8, (1, 9, 4), (1, 12, 4)), (11, 16, (1, 20, 8), (1, 16, 6))])
def test_fri_transcript_of_a_shard_proof_in_circuit(ctx, oracle, log_n, width, inner, outer):
    """a shard proof made here; its FRI view with the challenger's capacity; the wired machine whose Poseidon2 chip starts with the
    sponge chain over the layer roots: key and proof bytes against the oracle on the independently restated arrays, four verifiers;
    challenges that the transcript does not produce are refused before anything is proven"""
    from zktls_amd.device import fri_view_shard_paths, fri_view_transcript, verify_fri_transcript
    import poseidon2_air as P2
    O = oracle
    iprm, prm, oprm = Params(*inner), Params(*outer), O.default_params(*outer)
    pv = [4, 5]
    trace = ctx.gen_trace(SEED, 5, log_n, width)
    shard_proof = ctx.prove_shard(trace, log_n, width, pv, iprm)
    trace.free()
    view = fri_view_shard_paths(shard_proof, log_n, width, pv, iprm)
    roots, betas, capacity, pending = fri_view_transcript(shard_proof, log_n, width, pv, iprm)
    assert roots == view["roots"] and betas == view["betas"] and pending == 0
    traces, pre, progs, tables, pub = F.machine_layers(view, capacity=capacity)
    lns, ws, pws = shape_of(traces, pre)
    assert ws[0] == P2.WIDTH_T and ws[3] == 8 and pws == [0, 0, 8, 12] and len(pub) == 12
    key = ctx.fri_transcript_key(view, prm)
    assert key.root.tolist() == O.machine_setup(pre, lns, oprm).tolist()
    proof = ctx.prove_fri_transcript(key, view, capacity, prm)
    oproof = O.prove_machine_keyed(traces, pre, progs, tables, pub, oprm)
    assert proof.tobytes() == oproof.tobytes(), "transcript machine proof differs from the oracle's"
    nq = len(view["queries"])
    assert verify_fri_transcript(proof, view["final"], capacity, log_n, nq, key.root, prm) == (0, 0)        # no challenge is handed to the verifier
    assert verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, prm) == (0, 0)
    assert O.verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, oprm) == 0
    assert pyverify_chips.verify(proof.tobytes(), lns, ws, pub, outer[0], outer[1], outer[2], programs=progs, tables=tables, pre_widths=pws,
                                 pre_root=[int(v) for v in key.root]) is True
    # another capacity: the chain no longer produces the view's challenges -- refused by the prover, and by the verifier of the good proof
    other = list(capacity)
    other[5] = (other[5] + 1) % P
    with pytest.raises(Exception):
        ctx.prove_fri_transcript(key, view, other, prm)
    assert verify_fri_transcript(proof, view["final"], other, log_n, nq, key.root, prm)[0] == -6
    # challenges that the transcript does not produce: the key does not depend on them, the prover refuses
    bad = dict(view)
    bad["betas"] = [list(b) for b in view["betas"]]
    bad["betas"][1][2] = (bad["betas"][1][2] + 1) % P
    okey = ctx.fri_transcript_key(bad, prm)
    assert okey.root.tolist() == key.root.tolist()
    with pytest.raises(Exception):
        ctx.prove_fri_transcript(key, bad, capacity, prm)
    # a key made from other layer roots does not accept the proof
    bad2 = dict(view)
    bad2["roots"] = [list(r) for r in view["roots"]]
    bad2["roots"][0][0] = (bad2["roots"][0][0] + 1) % P
    okey2 = ctx.fri_transcript_key(bad2, prm)
    assert okey2.root.tolist() != key.root.tolist()
    assert verify_fri_transcript(proof, view["final"], capacity, log_n, nq, okey2.root, prm)[0] == -6
    key.close()
    okey.close()
    okey2.close()


def test_fullsize_fri_transcript_of_the_headline_shard(ctx, oracle):
    """the headline shard proof's 100 queries x 20 layers with the FRI transcript in-circuit: bytes against the oracle, timing"""
    from zktls_amd.device import fri_view_shard_paths, fri_view_transcript, verify_fri_transcript
    O = oracle
    log_n, width = 20, 256
    iprm, prm, oprm = Params(1, 100, 16), Params(1, 100, 16), O.default_params(1, 100, 16)
    trace = ctx.gen_trace(SEED, 32, log_n, width)
    shard_proof = ctx.prove_shard(trace, log_n, width, [1, 2, 3], iprm)
    trace.free()
    view = fri_view_shard_paths(shard_proof, log_n, width, [1, 2, 3], iprm)
    _, _, capacity, _ = fri_view_transcript(shard_proof, log_n, width, [1, 2, 3], iprm)
    key = ctx.fri_transcript_key(view, prm)
    proof = ctx.prove_fri_transcript(key, view, capacity, prm)
    t0 = time.perf_counter()
    proof = ctx.prove_fri_transcript(key, view, capacity, prm)
    t1 = time.perf_counter()
    assert verify_fri_transcript(proof, view["final"], capacity, log_n, 100, key.root, prm) == (0, 0)
    t2 = time.perf_counter()
    print("\nFRI layers + transcript of a 2^20 x 256 shard proof in-circuit: machine proof %.1f ms, %d bytes, host verification %.1f ms"
          % ((t1 - t0) * 1e3, proof.size, (t2 - t1) * 1e3))
    traces, pre, progs, tables, pub = F.machine_layers(view, capacity=capacity)
    lns, ws, pws = shape_of(traces, pre)
    assert proof.tobytes() == O.prove_machine_keyed(traces, pre, progs, tables, pub, oprm).tobytes()
    assert O.verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, oprm) == 0
    key.close()


# ------------------------------------------------------------------ the query-phase machine: proof of work and query indices in-circuit
@pytest.mark.parametrize("log_n,width,inner,outer", [(6, 8, (1, 9, 4), (1, 12, 4)), (11, 16, (1, 20, 8), (1, 16, 6)), (8, 8, (1, 40, 0), (1, 10, 4)),
                                                     (6, 8, (1, 300, 3), (1, 8, 2))])       # (300 queries: 38 sponge rows, the SAMPLES chip taller than the ROOTS table needs to be)
def test_fri_indices_of_a_shard_proof_in_circuit(ctx, oracle, log_n, width, inner, outer):
    """a shard proof made here; the machine whose sponge chain runs on through the final value and the proof-of-work witness and whose
    SAMPLES chip takes the bits of the words it then hands out: key (no index in it) and proof bytes against the oracle on the
    independently restated arrays, four verifiers; indices or a witness that the transcript does not produce are refused by the prover"""
    from zktls_amd.device import fri_view_shard_paths, fri_view_transcript, fri_view_witness, verify_fri_indices
    import poseidon2_air as P2
    O = oracle
    iprm, prm, oprm = Params(*inner), Params(*outer), O.default_params(*outer)
    pv = [4, 5]
    trace = ctx.gen_trace(SEED, 6, log_n, width)
    shard_proof = ctx.prove_shard(trace, log_n, width, pv, iprm)
    trace.free()
    view = fri_view_shard_paths(shard_proof, log_n, width, pv, iprm)
    _, _, capacity, _ = fri_view_transcript(shard_proof, log_n, width, pv, iprm)
    witness = fri_view_witness(shard_proof, log_n, width, pv, iprm)
    pow_bits, nq = inner[2], len(view["queries"])
    traces, pre, progs, tables, pub = F.machine_layers(view, capacity=capacity, query_phase=(witness, pow_bits))
    lns, ws, pws = shape_of(traces, pre)
    assert ws[0] == P2.WIDTH_T and ws[4] == F.S_MAIN and pws == [0, 0, 8, 12, F.S_PRE] and len(pub) == 12 and lns == sorted(lns, reverse=True)
    key = ctx.fri_indices_key(view, pow_bits, prm)
    assert key.root.tolist() == O.machine_setup(pre, lns, oprm).tolist()
    proof = ctx.prove_fri_indices(key, view, capacity, witness, pow_bits, prm)
    oproof = O.prove_machine_keyed(traces, pre, progs, tables, pub, oprm)
    assert proof.tobytes() == oproof.tobytes(), "query-phase machine proof differs from the oracle's"
    assert verify_fri_indices(proof, view["final"], capacity, log_n, nq, pow_bits, key.root, prm) == (0, 0)
    assert verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, prm) == (0, 0)
    assert O.verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, oprm) == 0
    assert pyverify_chips.verify(proof.tobytes(), lns, ws, pub, outer[0], outer[1], outer[2], programs=progs, tables=tables, pre_widths=pws,
                                 pre_root=[int(v) for v in key.root]) is True
    # the key does not depend on the indices: a view with another index has the same key, and the prover refuses it
    bad = dict(view)
    bad["queries"] = [tuple(q) for q in view["queries"]]
    i0, v0, s0 = bad["queries"][0]
    bad["queries"][0] = (i0 ^ 1, v0, s0)
    okey = ctx.fri_indices_key(bad, pow_bits, prm)
    assert okey.root.tolist() == key.root.tolist()
    with pytest.raises(Exception):
        ctx.prove_fri_indices(key, bad, capacity, witness, pow_bits, prm)
    # another witness: other words come out of the sponge
    with pytest.raises(Exception):
        ctx.prove_fri_indices(key, view, capacity, (witness + 1) % P, pow_bits, prm)
    # another capacity / final value / proof-of-work claim at the verifier
    other = list(capacity)
    other[5] = (other[5] + 1) % P
    assert verify_fri_indices(proof, view["final"], other, log_n, nq, pow_bits, key.root, prm)[0] == -6
    assert verify_fri_indices(proof, view["final"], capacity, log_n, nq, pow_bits + 1, key.root, prm)[0] == -6
    key.close()
    okey.close()


def test_fullsize_fri_indices_of_the_headline_shard(ctx, oracle):
    """the headline shard proof's 100 queries x 20 layers with the transcript's commit AND query phase in-circuit: bytes against the oracle, timing"""
    from zktls_amd.device import fri_view_shard_paths, fri_view_transcript, fri_view_witness, verify_fri_indices
    O = oracle
    log_n, width = 20, 256
    iprm, prm, oprm = Params(1, 100, 16), Params(1, 100, 16), O.default_params(1, 100, 16)
    trace = ctx.gen_trace(SEED, 33, log_n, width)
    shard_proof = ctx.prove_shard(trace, log_n, width, [1, 2, 3], iprm)
    trace.free()
    view = fri_view_shard_paths(shard_proof, log_n, width, [1, 2, 3], iprm)
    _, _, capacity, _ = fri_view_transcript(shard_proof, log_n, width, [1, 2, 3], iprm)
    witness = fri_view_witness(shard_proof, log_n, width, [1, 2, 3], iprm)
    key = ctx.fri_indices_key(view, 16, prm)
    proof = ctx.prove_fri_indices(key, view, capacity, witness, 16, prm)
    t0 = time.perf_counter()
    proof = ctx.prove_fri_indices(key, view, capacity, witness, 16, prm)
    t1 = time.perf_counter()
    assert verify_fri_indices(proof, view["final"], capacity, log_n, 100, 16, key.root, prm) == (0, 0)
    t2 = time.perf_counter()
    print("\nFRI layers + transcript (commit and query phase) of a 2^20 x 256 shard proof in-circuit: machine proof %.1f ms, %d bytes, host verification %.1f ms"
          % ((t1 - t0) * 1e3, proof.size, (t2 - t1) * 1e3))
    traces, pre, progs, tables, pub = F.machine_layers(view, capacity=capacity, query_phase=(witness, 16))
    lns, ws, pws = shape_of(traces, pre)
    assert proof.tobytes() == O.prove_machine_keyed(traces, pre, progs, tables, pub, oprm).tobytes()
    assert O.verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, oprm) == 0
    key.close()


@pytest.mark.parametrize("log_n,width,inner,outer,n", [(8, 8, (1, 12, 4), (1, 10, 4), 12), (11, 16, (1, 20, 8), (1, 16, 6), 5)])
def test_fri_indices_batch_equals_the_single_calls(ctx, oracle, log_n, width, inner, outer, n):
    """zkhip_prove_fri_indices_batch (lock-step lanes, and one context per worker): every proof the bytes of the step-by-step path
    (view -> key -> prove), every (vk, final value, capacity) what that path hands the verifier; a shard proof that does not verify fails its
    own job only"""
    from zktls_amd.device import (fri_view_shard_paths, fri_view_transcript, fri_view_witness, prove_fri_indices_batch, set_lockstep,
                                  verify_fri_indices)
    iprm, prm = Params(*inner), Params(*outer)
    shard_proofs, pvs = [], []
    for k in range(n):
        trace = ctx.gen_trace(SEED, 50 + k, log_n, width)
        pvs.append([k, 7])
        shard_proofs.append(ctx.prove_shard(trace, log_n, width, pvs[-1], iprm))
        trace.free()
    want = []
    for sp, pv in zip(shard_proofs, pvs):
        view = fri_view_shard_paths(sp, log_n, width, pv, iprm)
        _, _, capacity, _ = fri_view_transcript(sp, log_n, width, pv, iprm)
        witness = fri_view_witness(sp, log_n, width, pv, iprm)
        key = ctx.fri_indices_key(view, inner[2], prm)
        want.append((ctx.prove_fri_indices(key, view, capacity, witness, inner[2], prm).tobytes(), key.root.tolist(), view["final"], capacity))
        key.close()
    for batch in (16, 0):
        set_lockstep(batch)
        try:
            got = prove_fri_indices_batch(shard_proofs, log_n, width, pvs, iprm, prm, devices=[0], verify=True)
        finally:
            set_lockstep(16)
        assert len(got) == n
        for (proof, vk, final, cap), (wproof, wvk, wfinal, wcap) in zip(got, want):
            assert proof.tobytes() == wproof and vk == wvk and final == list(wfinal) and cap == list(wcap)
            assert verify_fri_indices(proof, final, cap, log_n, inner[1], inner[2], vk, prm) == (0, 0)
    # a corrupted shard proof: its job fails, the call reports it
    bad = [sp.copy() for sp in shard_proofs]
    bad[1][len(bad[1]) // 2] ^= 1
    with pytest.raises(Exception):
        prove_fri_indices_batch(bad, log_n, width, pvs, iprm, prm, devices=[0])
