"""RISC Zero Hal operator set: HIP kernels through the C ABI against the oracle's restatement
(oracle/hal.c, itself pinned against hashlib / first principles in tests/test_oracle_hal.py).  Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 2013265921


def test_eltwise_operators(ctx, oracle):
    rng = np.random.default_rng(11)
    for n in (1, 63, 1 << 16, (1 << 20) + 3):
        a, b = rng.integers(0, P, n, dtype=np.uint32), rng.integers(0, P, n, dtype=np.uint32)
        da, db = ctx.from_numpy(a), ctx.from_numpy(b)
        assert (ctx.eltwise_add(da, db).download() == oracle.hal_eltwise_add(a, b)).all()
        assert (ctx.eltwise_copy(da).download() == a).all()
    raw = rng.integers(0, P, 5000, dtype=np.uint32)
    raw[::13] = 0xFFFFFFFF
    d = ctx.from_raw(raw)
    assert (ctx.eltwise_zeroize(d).download_monty() == oracle.hal_eltwise_zeroize(raw)).all()
    for count, to_add in ((1, 1), (33, 5), (4096, 16), (100000, 3)):
        e = rng.integers(0, P, (to_add, count, 4), dtype=np.uint32)
        got = ctx.eltwise_sum_ext(ctx.from_numpy(e), count).download()
        assert (got == oracle.hal_eltwise_sum_ext(e, count)).all()


@pytest.mark.parametrize("count,log_size", [(1, 0), (3, 4), (7, 10), (2, 17)])
def test_zk_shift(ctx, oracle, count, log_size):
    rng = np.random.default_rng(count + log_size)
    polys = rng.integers(0, P, (count, 1 << log_size), dtype=np.uint32)
    for shift in (3, 31, P - 1):
        got = ctx.zk_shift(ctx.from_numpy(polys), count, log_size, shift).download()
        assert (got == oracle.hal_zk_shift(polys, count, log_size, shift)).all()


def test_zk_shift_of_a_slice_at_a_four_byte_offset(ctx, oracle):
    """a Hal slice inside a larger buffer need not be 16-byte aligned: the 4-bytes-per-lane form serves it"""
    import ctypes as C
    from zktls_amd.device import DeviceBuffer
    rng = np.random.default_rng(77)
    count, log_size = 3, 9
    polys = rng.integers(0, P, (count, 1 << log_size), dtype=np.uint32)
    big = ctx.alloc(polys.size + 8)
    for off in (1, 2, 3):
        big.upload(np.concatenate([np.zeros(off, dtype=np.uint32), polys.ravel(), np.zeros(8 - off, dtype=np.uint32)]))
        view = DeviceBuffer(ctx, polys.size, ptr=big.ptr + 4 * off, owner=big)
        ctx.zk_shift(view, count, log_size, 3)
        got = big.download()
        assert (got[off:off + polys.size] == np.asarray(oracle.hal_zk_shift(polys, count, log_size, 3)).ravel()).all() and not got[:off].any() and not got[off + polys.size:].any()


@pytest.mark.parametrize("ext_field", [0, 1])
def test_mix_poly_coeffs_and_batch_evaluate_any(ctx, oracle, ext_field):
    rng = np.random.default_rng(20 + ext_field)
    for count, input_size, ncombo in ((6, 5, 3), (1 << 12, 40, 4), ((1 << 16) + 5, 9, 2), (777, 150, 11), (64, 1, 1), (70, 4100, 3), (1 << 12, 333, 40)):      # (4100 inputs: beyond the plan kernel's LDS, the register form with per-term fallbacks; 40 combos in runs)
        inp = rng.integers(0, P, (input_size, count), dtype=np.uint32)
        combos = rng.integers(0, ncombo, input_size, dtype=np.uint32)
        start, mix = rng.integers(0, P, 4, dtype=np.uint32), rng.integers(0, P, 4, dtype=np.uint32)
        out0 = rng.integers(0, P, (ncombo, count, 4), dtype=np.uint32)
        d_out = ctx.from_numpy(out0)
        ctx.mix_poly_coeffs(d_out, start, mix, ctx.from_numpy(inp), ctx.from_raw(combos), input_size, count, ext_field)
        assert (d_out.download() == oracle.hal_mix_poly_coeffs(out0, start, mix, inp, combos, input_size, count, ext_field)).all()
    for npoly, log_size, nev in ((3, 0, 2), (3, 4, 5), (5, 11, 9), (2, 16, 3), (2, 10, 2), (3, 15, 4), (2, 18, 3)):
        polys = rng.integers(0, P, (npoly, 1 << log_size), dtype=np.uint32)
        which = rng.integers(0, npoly, nev, dtype=np.uint32)
        xs = rng.integers(0, P, (nev, 4), dtype=np.uint32)
        got = ctx.batch_evaluate_any(ctx.from_numpy(polys), log_size, ctx.from_raw(which), ctx.from_numpy(xs), ext_field).download()
        assert (got == oracle.hal_batch_evaluate_any(polys, log_size, which, xs, ext_field)).all()


@pytest.mark.parametrize("ext_field", [0, 1])
@pytest.mark.parametrize("n", [1, 7, 8, 9, 2048, 2049, 70000, (1 << 20) + 11])
def test_prefix_products_ext(ctx, oracle, ext_field, n):
    rng = np.random.default_rng(n)
    v = rng.integers(0, P, (n, 4), dtype=np.uint32)
    got = ctx.prefix_products_ext(ctx.from_numpy(v), ext_field).download()
    assert (got == oracle.hal_prefix_products_ext(v, ext_field)).all()


def test_gather_sample_and_scatter(ctx, oracle):
    rng = np.random.default_rng(31)
    size, stride = 300, 1 << 12
    src = rng.integers(0, P, (size, stride), dtype=np.uint32)
    d = ctx.from_numpy(src)
    for idx in (0, 17, stride - 1):
        assert (ctx.gather_sample(d, idx, size, stride).download() == oracle.hal_gather_sample(src, idx, size, stride)).all()
    rows = 500
    lens = rng.integers(0, 6, rows)
    index = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    total = int(index[-1])
    offsets = rng.permutation(20000)[:total].astype(np.uint32)
    values = rng.integers(0, P, total, dtype=np.uint32)
    into = np.zeros(20000, dtype=np.uint32)
    d_into = ctx.from_raw(into)
    ctx.scatter(d_into, ctx.from_raw(index), ctx.from_raw(offsets), ctx.from_raw(values))
    assert (d_into.download_monty() == oracle.hal_scatter(into, index, offsets, values)).all()


@pytest.mark.parametrize("cols,log_rows", [(1, 3), (13, 5), (14, 6), (16, 6), (45, 10), (256, 12)])
def test_sha256_hash_rows_fold_and_tree(ctx, oracle, cols, log_rows):
    rows = 1 << log_rows
    rng = np.random.default_rng(cols)
    m = rng.integers(0, P, (cols, rows), dtype=np.uint32)
    d = ctx.from_numpy(m)
    leaves = ctx.hash_rows_sha256(d, cols, rows).download_monty().reshape(rows, 8)
    want = oracle.hal_hash_rows_sha256(m)
    assert (leaves == want).all()
    tree = ctx.merkle_commit_sha256_colmajor(d, cols, log_rows).download_monty().reshape(-1, 8)
    level, off = want, 0
    assert (tree[:rows] == level).all()
    while level.shape[0] > 1:
        off += level.shape[0]
        level = oracle.hal_hash_fold_sha256(level)
        assert (tree[off:off + level.shape[0]] == level).all()
