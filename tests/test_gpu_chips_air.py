"""Chips with their own constraint programs on the GPU (zkhip_prove_chips_air, proof version 9): bytes against the oracle on mixed
sets, and a machine with the SHA-256 compression chip as one of its tables."""
import hashlib

import numpy as np
import pytest

import airs
import sha256_air as S
from zktls_amd._lib import Params, ZkHipError
from zktls_amd.device import sha256_air, verify_chips_air

pytestmark = pytest.mark.gpu
SEED = 0x5A4B544C53
P = 2013265921


@pytest.mark.parametrize("shape", [(1, 8, 4), (2, 6, 0), (3, 5, 2)])
def test_mixed_chip_set_bytes_equal_the_oracles(ctx, oracle, shape):
    O = oracle
    fib = airs.fibonacci_program()
    ft, pub = airs.fibonacci_trace(6, 3, 5)
    cnt = airs.counter_program(16).copy()
    cnt[4] = 3
    ct, _ = airs.counter_trace(10, 16, 3, 5)
    syn = O.gen_trace(SEED, 2, 9, 12)
    syn2 = O.gen_trace(SEED, 3, 9, 8)
    traces, progs = [ct, syn, syn2, ft], [cnt, None, None, fib]
    chips = [(ctx.from_numpy(t), t.shape[0].bit_length() - 1, t.shape[1]) for t in traces]
    proof = ctx.prove_chips_air(chips, progs, pub, Params(*shape))
    oproof = O.prove_chips_air(traces, progs, pub, O.default_params(*shape))
    assert proof.tobytes() == oproof.tobytes()
    log_ns, widths = [c[1] for c in chips], [c[2] for c in chips]
    assert verify_chips_air(proof, log_ns, widths, progs, pub, Params(*shape)) == (0, 0)
    assert O.verify_chips_air(proof, log_ns, widths, progs, pub, O.default_params(*shape)) == 0


def test_a_machine_with_the_sha256_chip(ctx, oracle):
    """the SHA-256 chip (2^10 rows x 640) next to a counter table and a synthetic table: one proof, bytes equal the oracle's"""
    O = oracle
    msg = bytes(range(200)) * 4                                    # 800 bytes -> 13 blocks -> 16 blocks
    sha_t, sha_pub = S.trace(S.pad(msg))
    assert S.digest_bytes(sha_pub) == hashlib.sha256(msg).digest()
    d_sha, limbs = ctx.sha256_gen_trace(S.pad(msg))
    assert limbs.tolist() == sha_pub
    cnt = airs.counter_program(8).copy()
    cnt[4] = S.N_PUBLIC                                            # over the shard's public values (the SHA-256 chip's 91): start = limb 0, step = limb 1
    ct, _ = airs.counter_trace(8, 8, sha_pub[0], sha_pub[1])
    syn = O.gen_trace(SEED, 5, 6, 4)
    progs = [sha256_air(), cnt, None]
    chips = [(d_sha, 10, 640), (ctx.from_numpy(ct), 8, 8), (ctx.from_numpy(syn), 6, 4)]
    proof = ctx.prove_chips_air(chips, progs, sha_pub, Params(1, 10, 4))
    oproof = O.prove_chips_air([sha_t, ct, syn], [S.program(), cnt, None], sha_pub, O.default_params(1, 10, 4))
    assert proof.tobytes() == oproof.tobytes()
    assert verify_chips_air(proof, [10, 8, 6], [640, 8, 4], progs, sha_pub, Params(1, 10, 4)) == (0, 0)
    wrong = list(sha_pub)
    wrong[7] ^= 1
    assert verify_chips_air(proof, [10, 8, 6], [640, 8, 4], progs, wrong, Params(1, 10, 4))[0] == -6


def test_misuse_fails_loudly(ctx, oracle):
    t = ctx.from_numpy(oracle.gen_trace(SEED, 1, 6, 4))
    with pytest.raises(ZkHipError):
        ctx.prove_chips_air([(t, 6, 4)], [airs.quintic_program()], [1], Params(1, 5, 3))          # degree 5 needs log_blowup >= 2
    with pytest.raises(ZkHipError):
        ctx.prove_chips_air([(t, 6, 4)], [airs.fibonacci_program()], [1, 2], Params(1, 5, 3))     # n_public mismatch


@pytest.mark.parametrize("shape", [(2, 6, 3), (3, 4, 0)])
def test_chips_of_degree_5_bytes_equal_the_oracles(ctx, oracle, shape):
    """a table with a degree-5 program (four quotient chunks, 16 quotient columns) between tables with two: version 9 with the program's
    log_quotient_degree in the header's has-program word"""
    O = oracle
    qt, qpub = airs.quintic_trace(9, 3)
    lin = O.air_program(4, 1, [(O.SEL_FIRST, [(1, [O.air_var(0)]), (P - 1, [O.air_var(0, public=True)])])])
    traces = [O.gen_trace(SEED, 7, 10, 8), qt, airs.quintic_trace(9, 3)[0], O.gen_trace(SEED, 1, 6, 4), np.full((32, 4), qpub[0], dtype=np.uint32)]
    progs = [None, airs.quintic_program(), airs.quintic_program(), None, lin]
    lns, ws = [t.shape[0].bit_length() - 1 for t in traces], [t.shape[1] for t in traces]
    chips = [(ctx.from_numpy(t), ln, w) for t, ln, w in zip(traces, lns, ws)]
    proof = ctx.prove_chips_air(chips, progs, qpub, Params(*shape))
    assert proof.tobytes() == O.prove_chips_air(traces, progs, qpub, O.default_params(*shape)).tobytes()
    assert verify_chips_air(proof, lns, ws, progs, qpub, Params(*shape)) == (0, 0)


@pytest.mark.parametrize("seed", range(6))
def test_machines_with_a_degree_5_table_bytes_equal_the_oracles(ctx, oracle, seed):
    """random machines (lookups as data) in which one table's program gains a degree-5 constraint (b^5 = b on its bit column): the lookup
    constraints of that table fold onto a quotient domain of four cosets; plain and keyed"""
    import machines as M
    from zktls_amd.device import verify_machine, verify_machine_keyed
    O = oracle
    V = O.air_var
    shape = [(2, 6, 3), (3, 4, 0)][seed % 2]
    traces, pre, progs, tables, pub = M.random_keyed_machine(500 + seed)
    full_w = [t.shape[1] + (0 if p is None else p.shape[1]) for t, p in zip(traces, pre)]
    k = seed % len(traces)
    w = full_w[k]
    progs = list(progs)
    progs[k] = O.air_program(w, 2, [(O.SEL_ALL, [(1, [V(w - 1), V(w - 1)]), (P - 1, [V(w - 1)])]),
                                    (O.SEL_ALL, [(1, [V(w - 1)] * 5), (P - 1, [V(w - 1)])])])
    assert O.air_log_quotient_degree(progs[k]) == 2
    lns, ws = [t.shape[0].bit_length() - 1 for t in traces], [t.shape[1] for t in traces]
    pws = [0 if p is None else p.shape[1] for p in pre]
    prm, oprm = Params(*shape), O.default_params(*shape)
    chips = [(ctx.from_numpy(t), ln, w_) for t, ln, w_ in zip(traces, lns, ws)]
    key = ctx.machine_setup([(None if p is None else ctx.from_numpy(p), ln, pw) for p, ln, pw in zip(pre, lns, pws)], prm)
    proof = ctx.prove_machine_keyed(key, chips, progs, tables, pub, prm)
    assert proof.tobytes() == O.prove_machine_keyed(traces, pre, progs, tables, pub, oprm).tobytes()
    assert verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, prm) == (0, 0)
    # the same tables as a plain machine (the preprocessed columns back in the main traces)
    whole = [t if p is None else np.ascontiguousarray(np.concatenate([p, t], axis=1)) for t, p in zip(traces, pre)]
    chips2 = [(ctx.from_numpy(t), ln, t.shape[1]) for t, ln in zip(whole, lns)]
    proof2 = ctx.prove_machine(chips2, progs, tables, pub, prm)
    assert proof2.tobytes() == O.prove_machine(whole, progs, tables, pub, oprm).tobytes()
    assert verify_machine(proof2, lns, full_w, progs, tables, pub, prm) == (0, 0)
