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
