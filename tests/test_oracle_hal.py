"""The oracle's restatement of the RISC Zero Hal operators (oracle/hal.c) against first principles: SHA-256 against hashlib,
extension products against pure-Python polynomial arithmetic, the rest against direct numpy / Python definitions."""
import hashlib
import struct

import numpy as np
import pytest

P = 2013265921


def py_ext_mul(a, b, w):
    t = [0] * 7
    for i in range(4):
        for j in range(4):
            t[i + j] = (t[i + j] + int(a[i]) * int(b[j])) % P
    return [(t[i] + (w * t[i + 4] if i + 4 < 7 else 0)) % P for i in range(4)]


@pytest.mark.parametrize("ext_field", [0, 1])
def test_ext_mul_both_fields(oracle, ext_field):
    rng = np.random.default_rng(1 + ext_field)
    w = 11 if ext_field == 0 else P - 11
    for _ in range(20):
        a, b = rng.integers(0, P, 4, dtype=np.uint32), rng.integers(0, P, 4, dtype=np.uint32)
        assert oracle.hal_ext_mul(a, b, ext_field).tolist() == py_ext_mul(a, b, w)
    # x * x^3 = x^4 = +-11
    assert oracle.hal_ext_mul([0, 1, 0, 0], [0, 0, 0, 1], ext_field).tolist() == [w, 0, 0, 0]


def test_sha256_rows_and_fold_against_hashlib(oracle):
    rng = np.random.default_rng(7)
    for cols in (1, 5, 13, 14, 15, 16, 17, 31, 32, 45):
        rows = 9
        m = rng.integers(0, P, (cols, rows), dtype=np.uint32)
        got = oracle.hal_hash_rows_sha256(m)
        for r in range(rows):
            want = hashlib.sha256(struct.pack(">%dI" % cols, *m[:, r].tolist())).digest()
            assert struct.pack(">8I", *got[r].tolist()) == want
    ch = rng.integers(0, 2**32, (6, 16), dtype=np.uint32)
    got = oracle.hal_hash_fold_sha256(ch)
    for i in range(6):
        assert struct.pack(">8I", *got[i].tolist()) == hashlib.sha256(struct.pack(">16I", *ch[i].tolist())).digest()


def test_elementwise_and_gather_scatter(oracle):
    rng = np.random.default_rng(3)
    a, b = rng.integers(0, P, 1000, dtype=np.uint32), rng.integers(0, P, 1000, dtype=np.uint32)
    assert (oracle.hal_eltwise_add(a, b) == ((a.astype(np.uint64) + b) % P)).all()
    z = a.copy(); z[::7] = 0xFFFFFFFF
    want = z.copy(); want[::7] = 0
    assert (oracle.hal_eltwise_zeroize(z) == want).all()
    e = rng.integers(0, P, (5, 33, 4), dtype=np.uint32)          # [to_add][count] extension elements
    assert (oracle.hal_eltwise_sum_ext(e, 33).reshape(33, 4) == (e.astype(np.uint64).sum(axis=0) % P)).all()
    src = rng.integers(0, P, (12, 64), dtype=np.uint32)          # column-major [size][stride]
    assert (oracle.hal_gather_sample(src, 17, 12, 64) == src[:, 17]).all()
    into = np.zeros(50, dtype=np.uint32)
    index, offsets, values = [0, 2, 2, 5], [7, 3, 40, 41, 9], [11, 12, 13, 14, 15]
    out = oracle.hal_scatter(into, index, offsets, values)
    assert out[[7, 3, 40, 41, 9]].tolist() == values and int(out.sum()) == sum(values)


def test_zk_shift_mix_evaluate_prefix(oracle):
    rng = np.random.default_rng(5)
    polys = rng.integers(0, P, (3, 16), dtype=np.uint32)
    sh = oracle.hal_zk_shift(polys, 3, 4, 3).reshape(3, 16)
    assert all(int(sh[p, i]) == int(polys[p, i]) * pow(3, i, P) % P for p in range(3) for i in range(16))
    for ext_field, w in ((0, 11), (1, P - 11)):
        # batch_evaluate_any by definition
        which = np.array([2, 0, 2], dtype=np.uint32)
        xs = rng.integers(0, P, (3, 4), dtype=np.uint32)
        got = oracle.hal_batch_evaluate_any(polys, 4, which, xs, ext_field).reshape(3, 4)
        for e in range(3):
            acc, xp = [0, 0, 0, 0], [1, 0, 0, 0]
            for i in range(16):
                acc = [(acc[k] + xp[k] * int(polys[which[e], i])) % P for k in range(4)]
                xp = py_ext_mul(xp, xs[e], w)
            assert got[e].tolist() == acc
        # prefix products
        v = rng.integers(0, P, (10, 4), dtype=np.uint32)
        pp = oracle.hal_prefix_products_ext(v, ext_field).reshape(10, 4)
        acc = [1, 0, 0, 0]
        for i in range(10):
            acc = py_ext_mul(acc, v[i], w)
            assert pp[i].tolist() == acc
        # mix_poly_coeffs
        count, input_size = 6, 5
        inp = rng.integers(0, P, (input_size, count), dtype=np.uint32)
        combos = np.array([0, 1, 0, 2, 1], dtype=np.uint32)
        start, mix = rng.integers(0, P, 4, dtype=np.uint32), rng.integers(0, P, 4, dtype=np.uint32)
        out0 = rng.integers(0, P, (3, count, 4), dtype=np.uint32)
        got = oracle.hal_mix_poly_coeffs(out0, start, mix, inp, combos, input_size, count, ext_field).reshape(3, count, 4)
        want = out0.astype(object).copy()
        for idx in range(count):
            cur = [int(x) for x in start]
            for i in range(input_size):
                for k in range(4):
                    want[combos[i], idx, k] = (int(want[combos[i], idx, k]) + cur[k] * int(inp[i, idx])) % P
                cur = py_ext_mul(cur, mix, w)
        assert (got == want.astype(np.uint32)).all()
