"""GPU parity tests for the STARK stages and the whole-shard prover: the HIP path through
the C ABI against the CPU oracle, bit-exact (proof BYTES included), plus the committed
golden proof fixtures and verifier acceptance."""
import hashlib
import json
import os

import numpy as np
import pytest

from zktls_amd._lib import Params
from zktls_amd.device import verify_shard

pytestmark = pytest.mark.gpu

P = 2013265921
SEED = 0x5A4B544C53
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "oracle_kat.json")))


@pytest.mark.parametrize("log_n,width", [(5, 4), (6, 8), (8, 12), (10, 16), (11, 36), (12, 256), (13, 260)])
def test_quotient_values_match_oracle(ctx, oracle, log_n, width):
    t = oracle.gen_trace(SEED, 2, log_n, width)
    lde = oracle.coset_lde(t, 1, 31)
    alpha = np.random.default_rng(log_n).integers(0, P, 4, dtype=np.uint32)
    exp = oracle.quotient_values(lde, log_n, alpha)
    got = ctx.quotient_values(ctx.from_numpy(lde), log_n, width, alpha).download().reshape(-1, 4)
    assert (got == exp).all()


@pytest.mark.parametrize("log_n,width", [(5, 4), (7, 8), (10, 20), (12, 64), (13, 3)])
def test_open_at_matches_oracle(ctx, oracle, log_n, width):
    m = oracle.fill_uniform(SEED + log_n, log_n, width)
    lde = oracle.coset_lde(m, 1, 31)
    rng = np.random.default_rng(log_n)
    z = rng.integers(0, P, (2, 4), dtype=np.uint32)
    d = ctx.from_numpy(lde)
    got = ctx.open_at(d, log_n, 1, width, z)
    assert (got[0] == oracle.open_at(lde, log_n, z[0])).all()
    assert (got[1] == oracle.open_at(lde, log_n, z[1])).all()
    one = ctx.open_at(d, log_n, 1, width, z[1:2])
    assert (one[0] == got[1]).all()


@pytest.mark.parametrize("log_h", [1, 2, 5, 10, 14, 17])
def test_fri_fold_matches_oracle(ctx, oracle, log_h):
    rng = np.random.default_rng(log_h)
    v = rng.integers(0, P, (1 << log_h, 4), dtype=np.uint32)
    beta = rng.integers(0, P, 4, dtype=np.uint32)
    got = ctx.fri_fold(ctx.from_numpy(v), log_h, beta).download().reshape(-1, 4)
    assert (got == oracle.fri_fold(v, beta)).all()


@pytest.mark.parametrize("log_n,width,q,pw,npub", [(5, 4, 4, 4, 0), (6, 8, 10, 8, 3), (9, 12, 20, 10, 1),
                                                    (10, 16, 100, 16, 3), (11, 24, 30, 12, 2), (12, 32, 100, 16, 3),
                                                    (14, 64, 100, 16, 0)])
def test_prove_shard_bytes_equal_oracle(ctx, oracle, log_n, width, q, pw, npub):
    pub = [7, 8, 9][:npub]
    trace = ctx.gen_trace(SEED, 3, log_n, width)
    prm = Params(1, q, pw)
    proof = ctx.prove_shard(trace, log_n, width, pub, prm)
    oprm = oracle.default_params(1, q, pw)
    oproof = oracle.prove_shard(oracle.gen_trace(SEED, 3, log_n, width), pub, oprm)
    dbg, odbg = ctx.prove_debug(), oracle.prove_debug()
    for k in ("trace_root", "alpha", "quotient_root", "zeta", "fri_alpha"):
        assert (dbg[k] == odbg[k]).all(), k
    assert dbg["pow_witness"] == odbg["pow_witness"]
    assert proof.size == oproof.size
    assert proof.tobytes() == oproof.tobytes()
    assert oracle.verify_shard(proof, log_n, width, pub, oprm) == 0
    assert verify_shard(proof, log_n, width, pub, prm) == (0, 0)


@pytest.mark.parametrize("name", sorted(KAT["proofs"]))
def test_golden_proofs_on_gpu(ctx, name):
    g = KAT["proofs"][name]
    trace = ctx.gen_trace(SEED, g["shard"], g["log_n"], g["width"])
    proof = ctx.prove_shard(trace, g["log_n"], g["width"], g["public"], Params(1, g["num_queries"], g["pow_bits"]))
    assert proof.size == g["bytes"]
    assert hashlib.sha256(proof.tobytes()).hexdigest() == g["sha256"]
    d = ctx.prove_debug()
    assert d["trace_root"].tolist() == g["trace_root"] and d["quotient_root"].tolist() == g["quotient_root"]
    assert d["pow_witness"] == g["pow_witness"]


@pytest.mark.parametrize("name", sorted(json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_kat.json")))["golden_proof_files"]))
def test_hip_prover_reproduces_the_committed_golden_proof_bytes(ctx, name):
    """tests/golden/proofs/*.bin (one per proof version; also checked on the CPU by three independent verifiers,
    tests/test_pyverify_cpu.py): the HIP prover must produce exactly these bytes -- no oracle involved at run time"""
    here = os.path.dirname(os.path.abspath(__file__))
    g = json.load(open(os.path.join(here, "golden", "oracle_kat.json")))["golden_proof_files"][name]
    want = open(os.path.join(here, "golden", "proofs", name + ".bin"), "rb").read()
    pairs = g["shape"][3]
    trace = (ctx.gen_trace_logup(g["seed"], g["shard"], g["log_n"], g["width"], pairs) if pairs
             else ctx.gen_trace(g["seed"], g["shard"], g["log_n"], g["width"]))
    proof = ctx.prove_shard(trace, g["log_n"], g["width"], g["public"], Params(*g["shape"]))
    assert proof.tobytes() == want


def test_invalid_trace_is_refused_or_rejected(ctx, oracle):
    log_n, width = 8, 8
    t = oracle.gen_trace(SEED, 0, log_n, width)
    t[17, 2] = (int(t[17, 2]) + 1) % P
    from zktls_amd._lib import ZkHipError
    prm = Params(1, 10, 4)
    try:
        proof = ctx.prove_shard(ctx.from_numpy(t), log_n, width, [], prm)
    except ZkHipError as e:
        assert e.code == -1
        return
    assert verify_shard(proof, log_n, width, [], prm)[0] == -6


def test_bad_arguments_fail_loudly(ctx):
    from zktls_amd._lib import ZkHipError
    trace = ctx.gen_trace(SEED, 0, 6, 8)
    with pytest.raises(ZkHipError):
        ctx.prove_shard(trace, 6, 6, [], Params(1, 10, 4))        # width not a multiple of 4
    with pytest.raises(ZkHipError):
        ctx.prove_shard(trace, 6, 8, [P], Params(1, 10, 4))       # non-canonical public value
    with pytest.raises(ZkHipError):
        ctx.coset_lde(trace, 23, 8)                               # log_n above the supported range (22)


# ------------------------------------------------------------------ LogUp
@pytest.mark.parametrize("log_n,width,pairs,shard", [(5, 8, 1, 0), (8, 16, 2, 1), (10, 24, 3, 2), (12, 256, 32, 3)])
def test_gen_trace_logup_and_perm_trace_match_oracle(ctx, oracle, log_n, width, pairs, shard):
    d = ctx.gen_trace_logup(SEED, shard, log_n, width, pairs)
    t = d.download().reshape(-1, width)
    assert (t == oracle.gen_trace_logup(SEED, shard, log_n, width, pairs)).all()
    rng = np.random.default_rng(log_n)
    gamma = rng.integers(0, P, 4, dtype=np.uint32)
    beta = rng.integers(0, P, 4, dtype=np.uint32)
    got = ctx.perm_trace(d, log_n, width, pairs, gamma, beta).download().reshape(-1, 4 * (pairs + 1))
    assert (got == oracle.perm_trace(t, pairs, gamma, beta)).all()
    assert got[-1, -4:].tolist() == [0, 0, 0, 0]


@pytest.mark.parametrize("log_n,width,pairs,q,pw", [(5, 8, 1, 4, 4), (6, 16, 2, 10, 8), (10, 24, 3, 30, 10), (12, 64, 8, 100, 16),
                                                    (13, 256, 32, 100, 16)])
def test_prove_shard_logup_bytes_equal_oracle(ctx, oracle, log_n, width, pairs, q, pw):
    trace = ctx.gen_trace_logup(SEED, 3, log_n, width, pairs)
    prm = Params(1, q, pw, pairs)
    proof = ctx.prove_shard(trace, log_n, width, [7, 8], prm)
    oprm = oracle.default_params(1, q, pw, pairs)
    oproof = oracle.prove_shard(oracle.gen_trace_logup(SEED, 3, log_n, width, pairs), [7, 8], oprm)
    assert proof.size == oproof.size
    assert proof.tobytes() == oproof.tobytes()
    assert oracle.verify_shard(proof, log_n, width, [7, 8], oprm) == 0
    assert verify_shard(proof, log_n, width, [7, 8], prm) == (0, 0)


def test_logup_rejects_trace_without_the_permutation(ctx):
    from zktls_amd._lib import ZkHipError
    log_n, width = 8, 16
    plain = ctx.gen_trace(SEED, 0, log_n, width)
    prm = Params(1, 10, 4, 2)
    try:
        proof = ctx.prove_shard(plain, log_n, width, [], prm)
    except ZkHipError as e:
        assert e.code == -1
        return
    assert verify_shard(proof, log_n, width, [], prm)[0] == -6


# ------------------------------------------------------------------ RISC0-style arity-16 fold (row a11)
@pytest.mark.parametrize("log_h,log_arity", [(4, 4), (6, 2), (10, 4), (12, 3), (16, 4)])
def test_fri_fold_k_matches_interpolation_definition(ctx, oracle, log_h, log_arity):
    rng = np.random.default_rng(log_h * 8 + log_arity)
    v = rng.integers(0, P, (1 << log_h, 4), dtype=np.uint32)
    beta = rng.integers(0, P, 4, dtype=np.uint32)
    got = ctx.fri_fold_k(ctx.from_numpy(v), log_h, log_arity, beta).download().reshape(-1, 4)
    assert (got == oracle.fri_fold_k(v, log_arity, beta)).all()


# ------------------------------------------------------------------ RISC-Zero-like shape (row a11) and the shapes between
# (log_blowup, queries, pow_bits, logup_pairs, log_fold, log_final, hash_width)
SHAPES = [
    (8, 8, (2, 10, 4, 0, 4, 0, 24)),
    (10, 16, (2, 20, 0, 0, 4, 2, 24)),
    (12, 16, (2, 50, 0, 0, 4, 8, 24)),        # RISC Zero's parameters at 2^12 rows
    (10, 32, (2, 20, 0, 2, 4, 6, 24)),        # with lookups
    (9, 8, (2, 10, 8, 0, 1, 0, 16)),          # only the blowup changes
    (9, 8, (1, 10, 8, 0, 3, 0, 16)),          # only the fold arity changes
    (9, 8, (1, 10, 8, 0, 1, 3, 24)),          # only final polynomial + hash change
    (10, 8, (3, 10, 0, 0, 2, 4, 16)),
    (14, 64, (2, 50, 0, 0, 4, 6, 24)),
    (13, 256, (2, 50, 0, 16, 4, 5, 24)),
    # code / data(/ accum) / check group order (8th field = code_width, proof version 8)
    (8, 16, (1, 10, 4, 0, 0, 0, 0, 4)),
    (10, 32, (2, 20, 0, 2, 4, 6, 24, 8)),
    (12, 64, (2, 50, 0, 0, 4, 8, 24, 16)),
    (13, 256, (2, 50, 0, 16, 4, 5, 24, 60)),  # data group width 196: the ragged tail of the width-24 sponge
    (11, 24, (1, 12, 6, 1, 1, 0, 16, 20)),
    (14, 128, (3, 16, 0, 0, 2, 4, 16, 124)),
]


@pytest.mark.parametrize("log_n,width,shape", SHAPES)
def test_prove_shard_other_shapes_bytes_equal_oracle(ctx, oracle, log_n, width, shape):
    pairs = shape[3]
    trace = ctx.gen_trace_logup(SEED, 5, log_n, width, pairs) if pairs else ctx.gen_trace(SEED, 5, log_n, width)
    otrace = oracle.gen_trace_logup(SEED, 5, log_n, width, pairs) if pairs else oracle.gen_trace(SEED, 5, log_n, width)
    prm, oprm = Params(*shape), oracle.default_params(*shape)
    proof = ctx.prove_shard(trace, log_n, width, [1, 2, 3], prm)
    oproof = oracle.prove_shard(otrace, [1, 2, 3], oprm)
    assert proof.size == oproof.size
    assert proof.tobytes() == oproof.tobytes()
    assert oracle.verify_shard(proof, log_n, width, [1, 2, 3], oprm) == 0
    assert verify_shard(proof, log_n, width, [1, 2, 3], prm) == (0, 0)


def test_shape_that_does_not_divide_is_refused(ctx):
    from zktls_amd._lib import ZkHipError
    trace = ctx.gen_trace(SEED, 0, 10, 8)
    with pytest.raises(ZkHipError):
        ctx.prove_shard(trace, 10, 8, [], Params(2, 10, 0, 0, 4, 0, 24))     # (10 - 0) % 4 != 0
    with pytest.raises(ZkHipError):
        ctx.prove_shard(trace, 10, 8, [], Params(2, 10, 0, 0, 4, 2, 20))     # hash width
    for cw in (8, 6, 12, -4):                                                  # code_width: a multiple of 4 below the width
        with pytest.raises(ZkHipError):
            ctx.prove_shard(trace, 10, 8, [], Params(2, 10, 0, 0, 4, 2, 24, cw))


@pytest.mark.parametrize("name", sorted(KAT["shape_proofs"]))
def test_golden_shape_proofs_on_gpu(ctx, name):
    g = KAT["shape_proofs"][name]
    pairs = g["shape"][3]
    trace = ctx.gen_trace_logup(SEED, g["shard"], g["log_n"], g["width"], pairs) if pairs else ctx.gen_trace(SEED, g["shard"], g["log_n"], g["width"])
    proof = ctx.prove_shard(trace, g["log_n"], g["width"], g["public"], Params(*g["shape"]))
    assert proof.size == g["bytes"]
    assert hashlib.sha256(proof.tobytes()).hexdigest() == g["sha256"]


@pytest.mark.parametrize("log_n,width,cw", [(8, 8, 0), (12, 32, 0), (14, 40, 0), (12, 32, 8), (14, 40, 12)])
def test_prove_segment_from_column_major_equals_oracle(ctx, oracle, log_n, width, cw):
    # RISC Zero's Hal layout: `width` contiguous columns; RISC-Zero-like shape; bytes equal the oracle's proof of the same trace
    from zktls_amd._lib import segment_params
    t = oracle.gen_trace(SEED, 9, log_n, width)
    cols = ctx.from_numpy(np.ascontiguousarray(t.T))
    lf = {8: 4, 12: 8, 14: 6}[log_n]
    proof = ctx.prove_segment(cols, log_n, width, [5], segment_params(50, 0, lf, cw))      # cw: code / data group commitments
    oproof = oracle.prove_shard(t, [5], oracle.segment_params(50, 0, lf, cw))
    assert proof.tobytes() == oproof.tobytes()
    assert verify_shard(proof, log_n, width, [5], segment_params(50, 0, lf, cw)) == (0, 0)


def test_concurrent_contexts_give_the_same_proofs(ctx, oracle):
    # bench.py keeps three shards in flight per GPU, each on its own context / HIP stream / host thread:
    # the proofs must not depend on what else is running
    import threading
    from zktls_amd.device import Context
    log_n, width, prm = 12, 32, Params(1, 30, 10)
    expected = []
    for s in range(3):
        expected.append(ctx.prove_shard(ctx.gen_trace(SEED, s, log_n, width), log_n, width, [s], prm).tobytes())
    ctxs = [Context(0) for _ in range(3)]
    got = [[None] * 4 for _ in range(3)]

    def worker(w):
        c = ctxs[w]
        t = c.gen_trace(SEED, w, log_n, width)
        for r in range(4):
            got[w][r] = c.prove_shard(t, log_n, width, [w], prm).tobytes()
    ts = [threading.Thread(target=worker, args=(w,)) for w in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for w in range(3):
        assert all(g == expected[w] for g in got[w])
        ctxs[w].close()


@pytest.mark.parametrize("log_n,width", [(8, 320), (9, 512), (8, 1024), (6, 36), (7, 100)])
def test_prove_shard_wide_and_ragged_widths(ctx, oracle, log_n, width):
    # widths that are not powers of two, and the widest supported trace (1024 columns: 16 column groups per lane in the quotient)
    prm, oprm = Params(1, 8, 4), oracle.default_params(1, 8, 4)
    proof = ctx.prove_shard(ctx.gen_trace(SEED, 2, log_n, width), log_n, width, [], prm)
    assert proof.tobytes() == oracle.prove_shard(oracle.gen_trace(SEED, 2, log_n, width), [], oprm).tobytes()
    assert verify_shard(proof, log_n, width, [], prm) == (0, 0)


def test_width_above_the_limit_is_refused(ctx):
    from zktls_amd._lib import ZkHipError
    with pytest.raises(ZkHipError):
        ctx.prove_shard(ctx.gen_trace(SEED, 0, 6, 1028), 6, 1028, [], Params(1, 8, 4))


# ------------------------------------------------------------------ shards of several chips with different heights (SP1's shard shape)
CHIP_SETS = [
    ([(8, 8)], (1, 10, 4)),
    ([(10, 16), (8, 8)], (1, 10, 4)),
    ([(10, 16), (10, 8), (7, 12), (7, 4), (5, 8)], (1, 20, 8)),
    ([(9, 8), (8, 8), (7, 8), (6, 8), (5, 8)], (2, 10, 0)),
    ([(11, 32), (6, 4)], (3, 8, 4)),
    ([(14, 64), (14, 16), (12, 40), (10, 256), (10, 8), (10, 8), (6, 4)], (1, 100, 16)),
    ([(16, 32), (13, 8), (5, 4)], (1, 30, 10)),
]


@pytest.mark.parametrize("chips,prm", CHIP_SETS)
def test_prove_chips_bytes_equal_oracle(ctx, oracle, chips, prm):
    from zktls_amd.device import verify_chips
    params, oparams = Params(*prm), oracle.default_params(*prm)
    dev = [(ctx.gen_trace(SEED, i, ln, w), ln, w) for i, (ln, w) in enumerate(chips)]
    host = [oracle.gen_trace(SEED, i, ln, w) for i, (ln, w) in enumerate(chips)]
    proof = ctx.prove_chips(dev, [3, 4], params)
    oproof = oracle.prove_chips(host, [3, 4], oparams)
    assert proof.size == oproof.size
    assert proof.tobytes() == oproof.tobytes()
    lns, ws = [c[0] for c in chips], [c[1] for c in chips]
    assert oracle.verify_chips(proof, lns, ws, [3, 4], oparams) == 0
    assert verify_chips(proof, lns, ws, [3, 4], params) == (0, 0)
    for b, _, _ in dev:
        b.free()


def test_prove_chips_rejects_bad_sets_and_bad_traces(ctx, oracle):
    from zktls_amd._lib import ZkHipError
    from zktls_amd.device import verify_chips
    a, b = ctx.gen_trace(SEED, 0, 6, 8), ctx.gen_trace(SEED, 1, 8, 8)
    with pytest.raises(ZkHipError):
        ctx.prove_chips([(a, 6, 8), (b, 8, 8)], [], Params(1, 4, 0))                 # not tallest first
    with pytest.raises(ZkHipError):
        ctx.prove_chips([(b, 8, 8), (a, 6, 8)], [], Params(1, 4, 0, 0, 4, 0, 24))    # only the SP1 FRI shape
    t = oracle.gen_trace(SEED, 1, 6, 8)
    t[9, 2] = (int(t[9, 2]) + 1) % P
    prm = Params(1, 6, 2)
    try:
        proof = ctx.prove_chips([(b, 8, 8), (ctx.from_numpy(t), 6, 8)], [], prm)
    except ZkHipError as e:
        assert e.code == -1
        return
    assert verify_chips(proof, [8, 6], [8, 8], [], prm) == (-6, 10)


@pytest.mark.parametrize("name", sorted(KAT["chip_proofs"]))
def test_golden_chip_proofs_on_gpu(ctx, name):
    g = KAT["chip_proofs"][name]
    dev = [(ctx.gen_trace(SEED, i, ln, w), ln, w) for i, (ln, w) in enumerate(g["chips"])]
    proof = ctx.prove_chips(dev, g["public"], Params(*g["params"]))
    assert proof.size == g["bytes"] and hashlib.sha256(proof.tobytes()).hexdigest() == g["sha256"]


def test_prove_shard_from_host_memory_equals_device_path(ctx, oracle):
    log_n, width = 11, 24
    t = oracle.gen_trace(SEED, 4, log_n, width)
    prm = Params(1, 20, 8)
    a = ctx.prove_shard_host(t, [6], prm)
    b = ctx.prove_shard(ctx.from_numpy(t), log_n, width, [6], prm)
    assert a.tobytes() == b.tobytes() == oracle.prove_shard(t, [6], oracle.default_params(1, 20, 8)).tobytes()


def _random_configs(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        log_n = int(rng.integers(5, 13))
        width = 4 * int(rng.integers(1, 19))
        b = int(rng.integers(1, 4))
        K = int(rng.integers(1, 5))
        fs = [f for f in range(0, min(log_n, 8) + 1) if (log_n - f) % K == 0]
        if not fs:
            continue
        F = int(rng.choice(fs))
        hw = int(rng.choice([16, 24]))
        pairs = int(rng.integers(0, width // 8 + 1)) if rng.random() < 0.4 else 0
        q, pw, npub = int(rng.integers(1, 13)), int(rng.integers(0, 7)), int(rng.integers(0, 6))
        out.append((log_n, width, (b, q, pw, pairs, K, F, hw), npub))
    return out


@pytest.mark.parametrize("log_n,width,shape,npub", _random_configs(40, 20261002))
def test_prove_shard_randomised_configurations(ctx, oracle, log_n, width, shape, npub):
    # a fixed pseudo-random sweep over sizes, widths, blowups, fold arities, final-polynomial lengths, hashes, lookups
    pairs = shape[3]
    pub = list(range(11, 11 + npub))
    trace = ctx.gen_trace_logup(SEED, 8, log_n, width, pairs) if pairs else ctx.gen_trace(SEED, 8, log_n, width)
    otrace = oracle.gen_trace_logup(SEED, 8, log_n, width, pairs) if pairs else oracle.gen_trace(SEED, 8, log_n, width)
    proof = ctx.prove_shard(trace, log_n, width, pub, Params(*shape))
    assert proof.tobytes() == oracle.prove_shard(otrace, pub, oracle.default_params(*shape)).tobytes()
    assert verify_shard(proof, log_n, width, pub, Params(*shape)) == (0, 0)
    trace.free()


def _random_chip_sets(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        n = int(rng.integers(1, 8))
        heights = sorted((int(h) for h in rng.integers(5, 12, size=n)), reverse=True)
        while max(heights.count(h) for h in heights) > 4:
            heights = sorted((int(h) for h in rng.integers(5, 12, size=n)), reverse=True)
        chips = []
        for h in heights:
            w = 4 * int(rng.integers(1, 12))
            chips.append((h, w, int(rng.integers(0, w // 8 + 1)) if rng.random() < 0.35 else 0))
        out.append((chips, (int(rng.integers(1, 4)), int(rng.integers(1, 10)), int(rng.integers(0, 6)))))
    return out


@pytest.mark.parametrize("chips,prm", _random_chip_sets(24, 777))
def test_prove_chips_randomised_sets(ctx, oracle, chips, prm):
    from zktls_amd.device import verify_chips
    dev = [(ctx.gen_trace_logup(SEED, 30 + i, ln, w, pr) if pr else ctx.gen_trace(SEED, 30 + i, ln, w), ln, w, pr) for i, (ln, w, pr) in enumerate(chips)]
    host = [oracle.gen_trace_logup(SEED, 30 + i, ln, w, pr) if pr else oracle.gen_trace(SEED, 30 + i, ln, w) for i, (ln, w, pr) in enumerate(chips)]
    pairs = [c[2] for c in chips]
    proof = ctx.prove_chips(dev, [5], Params(*prm))
    assert proof.tobytes() == oracle.prove_chips(host, [5], oracle.default_params(*prm), pairs).tobytes()
    assert verify_chips(proof, [c[0] for c in chips], [c[1] for c in chips], [5], Params(*prm), pairs) == (0, 0)
    for d in dev:
        d[0].free()


CHIP_SETS_LOGUP = [
    ([(8, 8, 1)], (1, 10, 4)),
    ([(10, 16, 2), (8, 8, 0)], (1, 10, 4)),
    ([(10, 16, 0), (8, 8, 1)], (1, 10, 4)),                      # the permutation tree is shorter than the trace tree
    ([(10, 16, 1), (10, 8, 1), (7, 24, 3), (7, 4, 0), (5, 8, 1)], (1, 20, 8)),
    ([(9, 8, 1), (8, 16, 2), (7, 8, 0)], (2, 10, 0)),
    ([(14, 64, 8), (12, 256, 32), (12, 8, 0), (9, 40, 5)], (1, 50, 12)),
]


@pytest.mark.parametrize("chips,prm", CHIP_SETS_LOGUP)
def test_prove_chips_with_lookups_bytes_equal_oracle(ctx, oracle, chips, prm):
    # chips with in-table LogUp pairs: their permutation traces form a third mixed-height commitment (proof version 5)
    from zktls_amd.device import verify_chips
    params, oparams = Params(*prm), oracle.default_params(*prm)
    dev, host = [], []
    for i, (ln, w, pr) in enumerate(chips):
        dev.append((ctx.gen_trace_logup(SEED, i, ln, w, pr) if pr else ctx.gen_trace(SEED, i, ln, w), ln, w, pr))
        host.append(oracle.gen_trace_logup(SEED, i, ln, w, pr) if pr else oracle.gen_trace(SEED, i, ln, w))
    pairs = [c[2] for c in chips]
    proof = ctx.prove_chips(dev, [3, 4], params)
    oproof = oracle.prove_chips(host, [3, 4], oparams, pairs)
    assert proof.tobytes() == oproof.tobytes()
    lns, ws = [c[0] for c in chips], [c[1] for c in chips]
    assert verify_chips(proof, lns, ws, [3, 4], params, pairs) == (0, 0)
    assert oracle.verify_chips(proof, lns, ws, [3, 4], oparams, pairs) == 0
    assert verify_chips(proof, lns, ws, [3, 4], params, None)[0] == -6          # not a proof of the lookup-free statement
    for d in dev:
        d[0].free()


CROSS_SETS = [
    # (log_n, width, pairs, partner)
    ([(9, 16, 2, 1), (9, 24, 2, 0), (7, 8, 1, -1), (6, 4, 0, -1)], (1, 12, 4)),
    ([(8, 8, 1, 1), (8, 8, 1, 0)], (1, 8, 0)),
    ([(11, 32, 0, -1), (10, 40, 3, 2), (10, 24, 3, 1), (10, 8, 1, -1), (6, 16, 2, 5), (6, 16, 2, 4)], (2, 20, 6)),
]


def _cross_traces(gen_cross, gen_logup, gen_plain, chips):
    out = []
    for i, (ln, w, pr, pa) in enumerate(chips):
        if pa >= 0:
            out.append(gen_cross(SEED, i, pa, ln, w, chips[pa][1], pr))
        elif pr:
            out.append(gen_logup(SEED, i, ln, w, pr))
        else:
            out.append(gen_plain(SEED, i, ln, w))
    return out


@pytest.mark.parametrize("chips,prm", CROSS_SETS)
def test_prove_chips_with_lookups_between_chips(ctx, oracle, chips, prm):
    # two chips of equal height hold each other's sender columns: every chip exposes the end of its running sum, the sums
    # cancel over the shard (sp1-stark's local cumulative sums; proof version 6)
    from zktls_amd.device import verify_chips
    params, oparams = Params(*prm), oracle.default_params(*prm)
    dtr = _cross_traces(ctx.gen_trace_logup_cross, ctx.gen_trace_logup, ctx.gen_trace, chips)
    htr = _cross_traces(oracle.gen_trace_logup_cross, oracle.gen_trace_logup, oracle.gen_trace, chips)
    for d, h in zip(dtr, htr):
        assert (d.download().reshape(h.shape) == h).all()
        assert oracle.check_trace(h) == 0
    lns, ws, prs, pas = ([c[k] for c in chips] for k in range(4))
    proof = ctx.prove_chips([(d, ln, w, pr, pa) for d, (ln, w, pr, pa) in zip(dtr, chips)], [9], params)
    assert proof.tobytes() == oracle.prove_chips(htr, [9], oparams, prs, pas).tobytes()
    assert verify_chips(proof, lns, ws, [9], params, prs, pas) == (0, 0)
    assert oracle.verify_chips(proof, lns, ws, [9], oparams, prs, pas) == 0
    for d in dtr:
        d.free()


def test_lookups_between_chips_must_balance(ctx, oracle):
    from zktls_amd._lib import ZkHipError
    from zktls_amd.device import verify_chips
    chips = [(8, 8, 1, 1), (8, 8, 1, 0)]
    prm = Params(1, 6, 0)
    good = _cross_traces(ctx.gen_trace_logup_cross, ctx.gen_trace_logup, ctx.gen_trace, chips)
    # chip 0 receives from a stream that is NOT chip 1's: the exposed sums no longer cancel (check 11)
    bad0 = ctx.gen_trace_logup_cross(SEED, 0, 5, 8, 8, 8, 1)
    try:
        proof = ctx.prove_chips([(bad0, 8, 8, 1, 1), (good[1], 8, 8, 1, 0)], [], prm)
    except ZkHipError as e:
        assert e.code == -1
    else:
        assert verify_chips(proof, [8, 8], [8, 8], [], prm, [1, 1], [1, 0]) == (-6, 11)
    # the honest cross traces claimed as lookups INSIDE each chip: the running sums do not end at zero (check 10)
    try:
        proof = ctx.prove_chips([(good[0], 8, 8, 1), (good[1], 8, 8, 1)], [], prm)
    except ZkHipError as e:
        assert e.code == -1
    else:
        assert verify_chips(proof, [8, 8], [8, 8], [], prm, [1, 1], None) == (-6, 10)
    with pytest.raises(ZkHipError):
        ctx.prove_chips([(good[0], 8, 8, 1, 1), (good[1], 8, 8, 1, -1)], [], prm)          # partnership must be mutual


@pytest.mark.parametrize("name", sorted(KAT["chip_lookup_proofs"]))
def test_golden_chip_lookup_proofs_on_gpu(ctx, name):
    g = KAT["chip_lookup_proofs"][name]
    chips = [tuple(c) for c in g["chips"]]
    dtr = _cross_traces(ctx.gen_trace_logup_cross, ctx.gen_trace_logup, ctx.gen_trace, chips)
    cross = any(c[3] >= 0 for c in chips)
    proof = ctx.prove_chips([(d, ln, w, pr) + ((pa,) if cross else ()) for d, (ln, w, pr, pa) in zip(dtr, chips)], g["public"], Params(*g["params"]))
    assert proof.size == g["bytes"] and hashlib.sha256(proof.tobytes()).hexdigest() == g["sha256"]


@pytest.mark.parametrize("in_flight", [1, 3, 8])
def test_prove_shards_batch_equals_one_by_one_and_the_oracle(ctx, oracle, in_flight):
    # zkhip_prove_shards: the shards of one execution, several in flight on internal contexts; bytes must not depend on that
    from zktls_amd.device import prove_shards
    log_n, width, prm = 9, 16, Params(1, 12, 6)
    traces = [ctx.gen_trace(SEED, s, log_n, width) for s in range(7)]
    pvs = [[5, 6, s] for s in range(7)]
    got = prove_shards(traces, log_n, width, pvs, prm, in_flight=in_flight)
    for s in range(7):
        one = ctx.prove_shard(traces[s], log_n, width, pvs[s], prm)
        assert got[s].tobytes() == one.tobytes()
        assert verify_shard(got[s], log_n, width, pvs[s], prm) == (0, 0)
    exp = oracle.prove_shard(oracle.gen_trace(SEED, 3, log_n, width), pvs[3], oracle.default_params(1, 12, 6))
    assert got[3].tobytes() == exp.tobytes()


def test_prove_shards_from_host_traces_and_failing_job(ctx, oracle):
    from zktls_amd.device import prove_shards
    from zktls_amd._lib import ZkHipError
    log_n, width, prm = 8, 8, Params(1, 10, 4)
    host = [oracle.gen_trace(SEED, s, log_n, width) for s in range(4)]
    got = prove_shards(host, log_n, width, [[s] for s in range(4)], prm, in_flight=2, host=True)
    for s in range(4):
        assert got[s].tobytes() == oracle.prove_shard(host[s], [s], oracle.default_params(1, 10, 4)).tobytes()
    with pytest.raises(ZkHipError):                       # a public value that is not canonical fails its job and the call
        prove_shards(host, log_n, width, [[0], [2013265921], [2], [3]], prm, in_flight=2, host=True)
