"""The AIR as data on the device: the quotient kernel that INTERPRETS a constraint program (stark.hip, quotient_air_kernel) and
whole proofs against programs (zkhip_prove_shard_air), byte for byte against the oracle (oracle/air.c)."""
import numpy as np
import pytest

import airs
import pyverify
from zktls_amd._lib import Params, ZkHipError
from zktls_amd.device import air_synthetic, verify_shard, verify_shard_air

pytestmark = pytest.mark.gpu
P = 2013265921
SEED = 0x5A4B544C53


@pytest.mark.parametrize("log_n,width", [(5, 4), (8, 8), (10, 24), (12, 64), (14, 256)])
def test_interpreted_synthetic_program_equals_the_specialised_kernel_and_the_oracle(ctx, oracle, log_n, width):
    prog = air_synthetic(width, 3)
    trace = ctx.gen_trace(SEED, 4, log_n, width)
    lde = ctx.coset_lde(trace, log_n, width)
    alpha = [11, 22, 33, 44]
    got = ctx.quotient_values_air(prog, lde, log_n, width, [1, 2, 3], alpha).download().reshape(-1, 4)
    assert (got == ctx.quotient_values(lde, log_n, width, alpha).download().reshape(-1, 4)).all()
    olde = lde.download().reshape(-1, width)
    assert (got == oracle.quotient_values_air(prog, olde, log_n, [1, 2, 3], alpha)).all()


@pytest.mark.parametrize("shape", [(1, 12, 6, 0, 0, 0, 0), (2, 8, 0, 0, 4, 2, 24), (3, 6, 4, 0, 1, 0, 16)])
def test_program_proofs_equal_the_oracles_bytes(ctx, oracle, shape):
    prm, oprm = Params(*shape), oracle.default_params(*shape)
    log_n = 10
    cases = []
    t, pub = airs.fibonacci_trace(log_n, 3, 5)
    cases.append((airs.fibonacci_program(), t, pub, 4))
    t, pub = airs.counter_trace(log_n, 16, 1234, 7)
    cases.append((airs.counter_program(16), t, pub, 16))
    cases.append((air_synthetic(32, 1), oracle.gen_trace(SEED, 2, log_n, 32), [9], 32))
    for prog, trace, pub, width in cases:
        proof = ctx.prove_shard_air(prog, ctx.from_numpy(trace), log_n, width, pub, prm)
        assert proof.tobytes() == oracle.prove_shard_air(prog, trace, pub, oprm).tobytes()
        assert verify_shard_air(prog, proof, log_n, width, pub, prm) == (0, 0)
        assert pyverify.verify(proof.tobytes(), log_n, width, pub, *shape, air=prog) is True
        assert verify_shard(proof, log_n, width, pub, prm)[0] == -6          # not a proof of the built-in AIR (version, digest)


@pytest.mark.parametrize("log_n,shape", [(8, (2, 8, 4, 0, 0, 0, 0)), (11, (2, 10, 0, 0, 1, 1, 24)), (12, (3, 6, 4, 0, 3, 0, 16))])
def test_degree_five_program_four_quotient_chunks(ctx, oracle, log_n, shape):
    """log_quotient_degree 2: quotient domain 4N, four chunks; quotient values and proof bytes equal the oracle's"""
    prog = airs.quintic_program()
    t, pub = airs.quintic_trace(log_n, 3)
    prm, oprm = Params(*shape), oracle.default_params(*shape)
    d = ctx.from_numpy(t)
    lde = ctx.coset_lde(d, log_n, 4, shape[0], 31)
    alpha = [5, 6, 7, 8]
    got = ctx.quotient_values_air(prog, lde, log_n, 4, pub, alpha, log_quotient_degree=2).download().reshape(-1, 4)
    assert (got == oracle.quotient_values_air(prog, lde.download().reshape(-1, 4), log_n, pub, alpha)).all()
    proof = ctx.prove_shard_air(prog, d, log_n, 4, pub, prm)
    assert proof.tobytes() == oracle.prove_shard_air(prog, t, pub, oprm).tobytes()
    assert verify_shard_air(prog, proof, log_n, 4, pub, prm) == (0, 0)
    assert pyverify.verify(proof.tobytes(), log_n, 4, pub, *shape, air=prog) is True
    with pytest.raises(ZkHipError):
        ctx.prove_shard_air(prog, d, log_n, 4, pub, Params(1, 8, 4))          # blowup 2 cannot hold the 4N-point quotient domain


@pytest.mark.parametrize("seed", range(8))
def test_random_programs(ctx, oracle, seed):
    """pseudo-random AIRs (tests/airs.py): degree, width, number of terms and the use of next-row / public variables vary"""
    rng = np.random.default_rng(1000 + seed)
    log_n, width, maxdeg = int(rng.integers(5, 12)), 4 * int(rng.integers(1, 7)), int(rng.choice([2, 3, 4, 5]))
    prog, trace, pub = airs.random_program_and_trace(seed, log_n, width, maxdeg)
    lqd = oracle.air_log_quotient_degree(prog)
    b = int(rng.integers(lqd, 4))
    shape = (b, int(rng.integers(2, 12)), int(rng.integers(0, 6)), 0, 1, 0, int(rng.choice([16, 24])))
    prm, oprm = Params(*shape), oracle.default_params(*shape)
    proof = ctx.prove_shard_air(prog, ctx.from_numpy(trace), log_n, width, pub, prm)
    assert proof.tobytes() == oracle.prove_shard_air(prog, trace, pub, oprm).tobytes()
    assert verify_shard_air(prog, proof, log_n, width, pub, prm) == (0, 0)
    if log_n <= 8:
        assert pyverify.verify(proof.tobytes(), log_n, width, pub, *shape, air=prog) is True


def test_program_proof_at_2_pow_18_rows(ctx, oracle):
    log_n, width = 18, 32
    prog = airs.counter_program(width)
    t, pub = airs.counter_trace(log_n, width, 5, 3)
    prm = Params(1, 40, 10)
    proof = ctx.prove_shard_air(prog, ctx.from_numpy(t), log_n, width, pub, prm)
    assert verify_shard_air(prog, proof, log_n, width, pub, prm) == (0, 0)
    assert oracle.verify_shard_air(prog, proof, log_n, width, pub, oracle.default_params(1, 40, 10)) == 0
    t[1000, 5] = (int(t[1000, 5]) + 1) % P                                  # one wrong cell: no accepted proof
    try:
        bad = ctx.prove_shard_air(prog, ctx.from_numpy(t), log_n, width, pub, prm)
    except ZkHipError:
        return
    assert verify_shard_air(prog, bad, log_n, width, pub, prm)[0] == -6


def test_bad_programs_fail_loudly(ctx):
    prog = airs.fibonacci_program()
    trace = ctx.alloc(4 << 6)
    with pytest.raises(ZkHipError):
        ctx.prove_shard_air(prog[:-1], trace, 6, 4, [1, 2, 3], Params(1, 6, 4))
    with pytest.raises(ZkHipError):
        ctx.prove_shard_air(prog, trace, 6, 4, [1, 2], Params(1, 6, 4))       # n_public does not match the program
    with pytest.raises(ZkHipError):
        ctx.prove_shard_air(prog, trace, 6, 4, [1, 2, 3], Params(1, 6, 4, 1))  # lookups belong to the built-in AIR


def test_many_public_values_take_the_row_per_lane_interpreter(ctx, oracle):
    """more than 64 public values do not fit the per-point LDS slots of the term-parallel kernel (stark.hip): the row-per-lane
    interpreter proves those programs -- same bytes as the oracle either way"""
    O = oracle
    V = O.air_var
    width, n_pub, log_n = 8, 80, 9
    cons = [(O.SEL_FIRST, [(1, [V(0)]), (P - 1, [V(79, public=True)])]),
            (O.SEL_TRANSITION, [(1, [V(0, True)]), (P - 1, [V(0)]), (P - 1, [V(3, public=True)])])]
    for j in range(1, width):
        cons.append((O.SEL_ALL, [(1, [V(j)]), (P - j, [V(0), V(0), V(70, public=True)]), (P - 1, [V(j - 1)])]))
    prog = O.air_program(width, n_pub, cons)
    pub = [(7 * i + 1) % P for i in range(n_pub)]
    n = 1 << log_n
    t = np.zeros((n, width), dtype=np.uint64)
    x = (pub[79] + pub[3] * np.arange(n, dtype=np.uint64)) % P
    t[:, 0] = x
    for j in range(1, width):
        t[:, j] = (x * x % P * pub[70] % P * j + t[:, j - 1]) % P
    t = t.astype(np.uint32)
    shape = (1, 9, 5)
    proof = ctx.prove_shard_air(prog, ctx.from_numpy(t), log_n, width, pub, Params(*shape))
    assert proof.tobytes() == O.prove_shard_air(prog, t, pub, O.default_params(*shape)).tobytes()
    assert verify_shard_air(prog, proof, log_n, width, pub, Params(*shape)) == (0, 0)


@pytest.mark.parametrize("n_monomials,lanes", [(300, 64), (3000, 128), (9000, 256)])
def test_term_kernel_with_64_128_and_256_lanes_per_group(ctx, oracle, n_monomials, lanes):
    """the term-parallel kernel takes one, two or four wavefronts per group of 8 points by the program's record count (one record per
    distinct monomial).  Programs with that many distinct monomials on any trace: constraint k = m_k - m_k over a monomial m_k of its own
    (the two terms merge into one record whose coefficient happens to be zero) next to a real counter constraint; quotient values
    against the oracle, which evaluates the program term by term"""
    import itertools
    O = oracle
    V = O.air_var
    width, log_n = 48, 7
    cons = [(O.SEL_FIRST, [(1, [V(0)]), (P - 1, [V(0, public=True)])]),
            (O.SEL_TRANSITION, [(1, [V(0, True)]), (P - 1, [V(0)]), (P - 1, [])])]
    for i, j, k in itertools.islice(itertools.combinations(range(width), 3), n_monomials):
        cons.append((O.SEL_ALL, [(5, [V(i), V(j), V(k, (i + j) % 3 == 0)]), (P - 5, [V(k, (i + j) % 3 == 0), V(i), V(j)])]))
    prog = O.air_program(width, 1, cons)
    rng = np.random.default_rng(n_monomials)
    n = 1 << log_n
    t = rng.integers(0, P, (n, width)).astype(np.uint64)
    t[:, 0] = (11 + np.arange(n)) % P
    t = t.astype(np.uint32)
    alpha = [3, 1, 4, 1]
    lde = ctx.coset_lde(ctx.from_numpy(t), log_n, width)
    got = ctx.quotient_values_air(prog, lde, log_n, width, [11], alpha).download().reshape(-1, 4)
    assert (got == O.quotient_values_air(prog, lde.download().reshape(-1, width), log_n, [11], alpha)).all()
    # and a whole proof through that path
    prm = Params(1, 6, 4)
    proof = ctx.prove_shard_air(prog, ctx.from_numpy(t), log_n, width, [11], prm)
    assert proof.tobytes() == O.prove_shard_air(prog, t, [11], O.default_params(1, 6, 4)).tobytes()
