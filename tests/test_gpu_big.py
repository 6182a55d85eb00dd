"""Transforms, LDEs and shard proofs of 2^21 and 2^22 rows (SP1 core shards reach those heights: reference benchmark.md:9;
SURVEY.md section 7 step 4).  Up to 2^20 rows a transform is two launches of the pass kernel; above that the rows split into
R = 2 / 4 classes, each class runs the 2^20-point machinery on its sub-matrix, and one streaming radix-R pass combines them.
Checked against the oracle at narrow widths (the oracle needs seconds there) and by round trips at full width."""
import numpy as np
import pytest

from zktls_amd._lib import Params, ZkHipError
from zktls_amd.device import verify_shard

pytestmark = pytest.mark.gpu
P = 2013265921
SEED = 0x5A4B544C53


@pytest.mark.parametrize("log_n,width", [(21, 4), (21, 12), (22, 4), (22, 7)])
def test_big_dft_matches_oracle(ctx, oracle, log_n, width):
    m = oracle.fill_uniform(SEED + log_n, log_n, width)
    d = ctx.fill_uniform(SEED + log_n, log_n, width)
    assert (d.download().reshape(-1, width) == m).all()
    exp = oracle.ntt(m)
    nat = ctx.dft(d, log_n, width)
    assert (nat.download().reshape(-1, width) == exp).all()
    rev = ctx.dft(d, log_n, width, bitrev_out=True).download().reshape(-1, width)
    idx = np.arange(1 << log_n, dtype=np.uint32)
    br = np.zeros_like(idx)
    for b in range(log_n):
        br |= ((idx >> b) & 1) << (log_n - 1 - b)
    assert (rev[br] == exp).all()
    back = ctx.dft(nat, log_n, width, inverse=True)
    assert (back.download().reshape(-1, width) == m).all()
    inv = ctx.dft(d, log_n, width, inverse=True)
    assert (inv.download().reshape(-1, width) == oracle.ntt(m, inverse=True)).all()
    # in place
    ctx.dft(d, log_n, width, out=d)
    assert (d.download().reshape(-1, width) == exp).all()
    for x in (d, nat, back, inv):
        x.free()


@pytest.mark.parametrize("log_n,width,log_blowup", [(21, 8, 1), (21, 5, 2), (22, 8, 1), (22, 4, 0)])
def test_big_coset_lde_matches_oracle(ctx, oracle, log_n, width, log_blowup):
    m = oracle.fill_uniform(SEED + 3 * log_n, log_n, width)
    d = ctx.from_numpy(m)
    lde = ctx.coset_lde(d, log_n, width, log_blowup, 31)
    assert (lde.download().reshape(-1, width) == oracle.coset_lde(m, log_blowup, 31)).all()
    d.free(); lde.free()


def test_big_lde_into_interleaved_columns(ctx, oracle):
    # the quotient chunks are extended into the two halves of an 8-column matrix (out_ld > width): padding columns stay untouched
    log_n = 21
    m = oracle.fill_uniform(SEED + 9, log_n, 4)
    d = ctx.from_numpy(m)
    out = ctx.from_numpy(np.full((2 << log_n, 8), 7, dtype=np.uint32))
    ctx.coset_lde(d, log_n, 4, 1, 31, out=out, out_ld=8, out_col=4)
    got = out.download().reshape(-1, 8)
    assert (got[:, 4:] == oracle.coset_lde(m, 1, 31)).all() and (got[:, :4] == 7).all()
    d.free(); out.free()


def test_big_lde_at_blowup_2_takes_any_pitch(ctx, oracle):
    # at blowup 2 the tile passes run on dense 2^20-row classes: a 2^22-row matrix may sit in rows 300 words apart (the 256-word limit of the
    # other blowups does not apply); eight columns into a pitch of 300, offset 40, everything else untouched
    # (checked on three windows of 2^17 rows -- the first, one across the middle, the last -- against the oracle's rows; the surrounding columns against what
    # they held before: the 10 GB of the whole buffer never cross the PCIe link or numpy)
    from zktls_amd._lib import from_monty
    log_n, pitch, win = 22, 300, 1 << 17
    m = oracle.fill_uniform(SEED + 11, log_n, 8)
    want = oracle.coset_lde(m, 1, 31)
    d = ctx.from_numpy(m)
    out = ctx.fill_uniform(SEED + 12, log_n, 2 * pitch)            # 2^23 rows of 300 words
    rows = 2 << log_n
    starts = (0, rows // 2 - win // 2, rows - win)
    before = [out.download_monty(win * pitch, r0 * pitch).reshape(win, pitch) for r0 in starts]
    ctx.coset_lde(d, log_n, 8, 1, 31, out=out, out_ld=pitch, out_col=40)
    for r0, was in zip(starts, before):
        got = out.download_monty(win * pitch, r0 * pitch).reshape(win, pitch)
        assert (from_monty(np.ascontiguousarray(got[:, 40:48])) == want[r0:r0 + win]).all(), r0
        assert (got[:, :40] == was[:, :40]).all() and (got[:, 48:] == was[:, 48:]).all(), r0
    d.free(); out.free()


@pytest.mark.parametrize("log_n,width", [(21, 512), (22, 256)])
def test_big_fullwidth_round_trip(ctx, log_n, width):
    src = ctx.fill_uniform(SEED + 77, log_n, width)
    fwd = ctx.dft(src, log_n, width, bitrev_out=True)
    a = src.download_monty(1 << 22)
    assert not (fwd.download_monty(1 << 22) == a).all()
    # undo the bit reversal through a second forward/inverse pair: inverse(forward_natural(x)) == x
    nat = ctx.dft(src, log_n, width, out=fwd)
    back = ctx.dft(nat, log_n, width, inverse=True, out=nat)
    n = width << log_n
    for off in (0, n // 2 - 12345, n - (1 << 22)):
        assert (back.download_monty(1 << 22, off) == src.download_monty(1 << 22, off)).all()
    src.free(); fwd.free()


def test_too_wide_big_matrix_is_refused(ctx):
    d = ctx.alloc(1024 << 21)
    with pytest.raises(ZkHipError):
        ctx.dft(d, 21, 1024)
    d.free()


@pytest.mark.parametrize("log_n,width,shape", [(21, 16, (1, 30, 8)), (22, 8, (1, 20, 8)), (21, 8, (2, 20, 0, 0, 4, 1, 24))])
def test_big_shard_proof_bytes_equal_the_oracles(ctx, oracle, log_n, width, shape):
    prm, oprm = Params(*shape), oracle.default_params(*shape)
    trace = ctx.gen_trace(SEED, 3, log_n, width)
    proof = ctx.prove_shard(trace, log_n, width, [4, 5], prm)
    op = oracle.prove_shard(oracle.gen_trace(SEED, 3, log_n, width), [4, 5], oprm)
    assert proof.tobytes() == op.tobytes()
    assert verify_shard(proof, log_n, width, [4, 5], prm) == (0, 0)
    trace.free()


def test_big_fullwidth_shard_proof_verifies(ctx, oracle):
    # a 2^21 x 256 shard (4 GiB LDE, 2^22-leaf trees): accepted by both verifiers, deterministic
    log_n, width = 21, 256
    prm = Params(1, 100, 16)
    trace = ctx.gen_trace(SEED, 11, log_n, width)
    proof = ctx.prove_shard(trace, log_n, width, [1, 2, 3], prm)
    assert verify_shard(proof, log_n, width, [1, 2, 3], prm) == (0, 0)
    assert oracle.verify_shard(proof, log_n, width, [1, 2, 3], oracle.default_params(1, 100, 16)) == 0
    bad = proof.copy().view(np.uint32)
    bad[-3] = (int(bad[-3]) + 1) % P
    assert verify_shard(bad.view(np.uint8), log_n, width, [1, 2, 3], prm)[0] == -6
    assert ctx.prove_shard(trace, log_n, width, [1, 2, 3], prm).tobytes() == proof.tobytes()
    trace.free()


def test_big_multichip_shard_bytes_equal_the_oracles(ctx, oracle):
    """chips above 2^20 rows in a multi-chip shard, as SP1's tallest chips are (benchmark.md:9: shards of 2^21 - 2^22 rows): a 2^22-row and
    a 2^21-row table next to small ones, proof bytes against the oracle"""
    from zktls_amd.device import verify_chips
    shapes = [(22, 8), (21, 16), (18, 32), (12, 4)]
    prm, oprm = Params(1, 20, 8), oracle.default_params(1, 20, 8)
    host = [oracle.gen_trace(SEED, 20 + i, ln, w) for i, (ln, w) in enumerate(shapes)]
    chips = [(ctx.from_numpy(t), ln, w) for t, (ln, w) in zip(host, shapes)]
    proof = ctx.prove_chips(chips, [7, 8], prm)
    assert proof.tobytes() == oracle.prove_chips(host, [7, 8], oprm).tobytes()
    assert verify_chips(proof, [s[0] for s in shapes], [s[1] for s in shapes], [7, 8], prm) == (0, 0)
    for c in chips:
        c[0].free()


def test_big_machine_with_lookups_bytes_equal_the_oracles(ctx, oracle):
    """the range machine with 2^21 users (its PICK table has 2^22 rows): interaction tables, permutation traces and cumulative sums above
    2^20 rows"""
    import machines as M
    from zktls_amd.device import verify_machine
    import os
    prev = min(8, os.cpu_count() or 1)
    oracle.set_threads(min(os.cpu_count() or 1, 96))
    try:
        traces, progs, tables, pub = M.range_machine(10, 21)
        lns, ws = [t.shape[0].bit_length() - 1 for t in traces], [t.shape[1] for t in traces]
        assert lns == [22, 21, 10]
        prm, oprm = Params(1, 16, 6), oracle.default_params(1, 16, 6)
        chips = [(ctx.from_numpy(t), ln, w) for t, ln, w in zip(traces, lns, ws)]
        proof = ctx.prove_machine(chips, progs, tables, pub, prm)
        assert proof.tobytes() == oracle.prove_machine(traces, progs, tables, pub, oprm).tobytes()
        assert verify_machine(proof, lns, ws, progs, tables, pub, prm) == (0, 0)
        for c in chips:
            c[0].free()
    finally:
        oracle.set_threads(prev)


@pytest.mark.parametrize("size", [(21, 4), (12, 11)])
def test_big_keyed_machine_bytes_equal_the_oracles(ctx, oracle, size):
    """keyed machines above 2^20 rows: 2^21 users of a small preprocessed byte table, and a 2^22-row PREPROCESSED table (setup commits a
    2^23-row LDE); keys and proof bytes against the oracle"""
    import machines as M
    from zktls_amd.device import verify_machine_keyed
    import os
    prev = min(8, os.cpu_count() or 1)
    oracle.set_threads(min(os.cpu_count() or 1, 96))
    try:
        traces, pre, progs, tables, pub = M.byte_machine(*size)
        lns, ws = [t.shape[0].bit_length() - 1 for t in traces], [t.shape[1] for t in traces]
        pws = [0 if p is None else p.shape[1] for p in pre]
        assert max(lns) == max(size[0], 2 * size[1]) >= 21
        prm, oprm = Params(1, 16, 6), oracle.default_params(1, 16, 6)
        chips = [(ctx.from_numpy(t), ln, w) for t, ln, w in zip(traces, lns, ws)]
        key = ctx.machine_setup([(None if p is None else ctx.from_numpy(p), ln, pw) for p, ln, pw in zip(pre, lns, pws)], prm)
        assert key.root.tolist() == oracle.machine_setup(pre, lns, oprm).tolist()
        proof = ctx.prove_machine_keyed(key, chips, progs, tables, pub, prm)
        assert proof.tobytes() == oracle.prove_machine_keyed(traces, pre, progs, tables, pub, oprm).tobytes()
        assert verify_machine_keyed(proof, lns, ws, pws, key.root, progs, tables, pub, prm) == (0, 0)
        key.close()
        for c in chips:
            c[0].free()
    finally:
        oracle.set_threads(prev)
