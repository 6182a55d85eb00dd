"""The FRI-only recursion mode on the GPU (zkhip_prove_fri_indices[_batch]; SURVEY.md 8f-4, second half): a shard is proven on the device, and a second
proof -- a keyed machine of the Poseidon2 chip (Merkle paths + transcript), the FRI-fold chip, the SAMPLES chip and two tables -- states that the FRI
check of that shard proof passes: paths, folds, challenges, proof of work and query indices in-circuit.  The machine proof must equal, byte for byte,
what the oracle's generic keyed-machine prover makes of the Python restatement's arrays (tests/fri_air.py, tests/poseidon2_air.py).
(Round 6 removed the three generations before it -- zkhip_prove_fri_queries / _layers / _transcript --; their chips are covered as parts of this machine and
of the shard verifier machines, their programs and views by tests/test_fri_chip_cpu.py.)"""
import time

import numpy as np
import pytest

import fri_air as F
import pyverify
import pyverify_chips
from zktls_amd._lib import Params
from zktls_amd.device import fri_view_shard, verify_machine_keyed, verify_shard

pytestmark = pytest.mark.gpu
SEED = 0x5A4B544C53
P = 2013265921


def shape_of(traces, pre):
    return ([t.shape[0].bit_length() - 1 for t in traces], [t.shape[1] for t in traces], [0 if p is None else p.shape[1] for p in pre])


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
