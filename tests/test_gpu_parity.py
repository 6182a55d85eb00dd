"""GPU parity tests: the HIP path, called through the C ABI (libzkhip.so), against the CPU
oracle on the same seeded inputs -- bit-exact (integer field arithmetic).  Sizes are the
ones the oracle finishes in seconds; full-size (2^20 x 256) checks go through
size-independent properties (round trips, linearity, checksums) in test_gpu_fullsize.py.
"""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 2013265921
SEED = 0x5A4B544C53
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "oracle_kat.json")))


def bitrev_perm(log_n):
    n = 1 << log_n
    idx = np.arange(n, dtype=np.uint32)
    out = np.zeros(n, dtype=np.uint32)
    for b in range(log_n):
        out |= ((idx >> b) & 1) << (log_n - 1 - b)
    return out


def test_loaded_library_is_the_hip_build(ctx):
    import zktls_amd._lib as L
    assert os.path.exists(L.LIB_PATH)
    assert L.device_count() >= 1


def test_montgomery_conversion_on_device(ctx):
    rng = np.random.default_rng(0)
    x = rng.integers(0, P, 10000, dtype=np.uint32)
    buf = ctx.from_numpy(x)                       # uploaded in Montgomery form by numpy
    import ctypes as C
    from zktls_amd._lib import check
    out = ctx.alloc(x.size)
    check(ctx.lib.zkhip_from_monty(ctx.handle, C.c_void_p(buf.ptr), C.c_void_p(out.ptr), x.size))
    assert (out.download_monty() == x).all()     # device from_monty gives canonical words
    check(ctx.lib.zkhip_to_monty(ctx.handle, C.c_void_p(out.ptr), C.c_void_p(out.ptr), x.size))
    assert (out.download() == x).all()


def test_fill_uniform_matches_oracle(ctx, oracle):
    for log_n, w in ((6, 4), (10, 24)):
        d = ctx.fill_uniform(SEED, log_n, w)
        assert (d.download().reshape(-1, w) == oracle.fill_uniform(SEED, log_n, w)).all()


def test_gen_trace_matches_oracle(ctx, oracle):
    for log_n, w, shard in ((6, 8, 0), (10, 16, 3), (12, 36, 7)):
        d = ctx.gen_trace(SEED, shard, log_n, w)
        t = d.download().reshape(-1, w)
        assert (t == oracle.gen_trace(SEED, shard, log_n, w)).all()
        assert oracle.check_trace(t) == 0


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16])
@pytest.mark.parametrize("width", [1, 3, 16])
def test_forward_dft_matches_oracle(ctx, oracle, log_n, width):
    m = oracle.fill_uniform(SEED + log_n, log_n, width)
    exp = oracle.ntt(m)
    src = ctx.from_numpy(m)
    nat = ctx.dft(src, log_n, width).download().reshape(-1, width)
    assert (nat == exp).all()
    br = ctx.dft(src, log_n, width, bitrev_out=True).download().reshape(-1, width)
    assert (br[bitrev_perm(log_n)] == exp).all()
    assert (src.download().reshape(-1, width) == m).all()      # input preserved


@pytest.mark.parametrize("log_n,width", [(0, 3), (1, 2), (3, 5), (4, 1), (5, 2), (8, 5), (10, 16), (11, 4), (13, 8), (15, 17), (18, 3)])
def test_inverse_dft_round_trip_and_oracle(ctx, oracle, log_n, width):
    m = oracle.fill_uniform(SEED + 100 + log_n, log_n, width)
    src = ctx.from_numpy(m)
    inv = ctx.dft(src, log_n, width, inverse=True)
    got = inv.download().reshape(-1, width)
    if log_n <= 15:
        assert (got == oracle.ntt(m, inverse=True)).all()
    back = ctx.dft(inv, log_n, width).download().reshape(-1, width)
    assert (back == m).all()


@pytest.mark.parametrize("log_n,width,log_blowup,shift", [
    (0, 2, 1, 31), (1, 3, 2, 31), (2, 4, 1, 5), (4, 5, 3, 31), (5, 4, 1, 31), (6, 4, 1, 31), (8, 7, 2, 31), (10, 16, 1, 31), (11, 8, 1, 31),
    (12, 24, 1, 31), (12, 4, 2, 5), (14, 33, 1, 31), (16, 16, 1, 31), (13, 8, 3, 31),
])
def test_coset_lde_matches_oracle(ctx, oracle, log_n, width, log_blowup, shift):
    m = oracle.fill_uniform(SEED + 7 * log_n + width, log_n, width)
    exp = oracle.coset_lde(m, log_blowup, shift)
    got = ctx.coset_lde(ctx.from_numpy(m), log_n, width, log_blowup, shift).download().reshape(-1, width)
    assert (got == exp).all()


def test_coset_lde_golden_fixtures(ctx):
    d = ctx.fill_uniform(SEED, 6, 4)
    lde = ctx.coset_lde(d, 6, 4)
    assert hashlib.sha256(lde.download().tobytes()).hexdigest() == KAT["lde_6x4_sha256"]
    tree = ctx.merkle_commit([(lde, 4)], 7)
    assert tree.download(8, offset=8 * ((2 << 7) - 2)).tolist() == KAT["lde_6x4_root"]
    d = ctx.fill_uniform(SEED + 1, 12, 24)
    lde = ctx.coset_lde(d, 12, 24)
    assert hashlib.sha256(lde.download().tobytes()).hexdigest() == KAT["lde_12x24_sha256"]
    tree = ctx.merkle_commit([(lde, 24)], 13)
    assert tree.download(8, offset=8 * ((2 << 13) - 2)).tolist() == KAT["lde_12x24_root"]


def test_coset_lde_2pow20_narrow_matches_oracle(ctx, oracle):
    # the headline row count, at a width the oracle still does in seconds
    log_n, width = 20, 4
    m = oracle.fill_uniform(SEED + 20, log_n, width)
    exp = oracle.coset_lde(m, 1, 31)
    got = ctx.coset_lde(ctx.from_numpy(m), log_n, width).download().reshape(-1, width)
    assert (got == exp).all()


def test_coset_lde_and_dft_2pow20_on_the_wide_tile_kernel(ctx, oracle):
    # 2^20 rows x 32 columns takes the 1024-row, two-columns-per-lane tile kernel of the headline shape, including its
    # rotated tile order on the strided passes (ntt.hip, launch_ntt_pass), and is still small enough for the oracle
    log_n, width = 20, 32
    m = oracle.fill_uniform(SEED + 21, log_n, width)
    d = ctx.from_numpy(m)
    got = ctx.coset_lde(d, log_n, width).download().reshape(-1, width)
    assert (got == oracle.coset_lde(m, 1, 31)).all()
    exp = oracle.ntt(m)
    assert (ctx.dft(d, log_n, width).download().reshape(-1, width) == exp).all()
    br = ctx.dft(d, log_n, width, bitrev_out=True).download().reshape(-1, width)
    assert (br[bitrev_perm(log_n)] == exp).all()


@pytest.mark.parametrize("width,log_blowup,out_ld", [(32, 1, None), (64, 1, 96), (96, 2, None)])
def test_fused_lde_equals_the_four_pass_sequence(ctx, oracle, width, log_blowup, out_ld):
    """2^20 rows x a multiple of 32 columns takes the fused middle launch (ntt_fused.hip: second inverse pass + first forward
    pass of two cosets in one kernel, coefficients never written); zkhip_ctx_set_lde_fusion(0) forces the unfused sequence.
    Both must give the same matrix bit for bit -- also with a padded output pitch and with four cosets (two fused launches) --
    and (narrowest case) the oracle's."""
    log_n = 20
    src = ctx.fill_uniform(SEED + 300 + width, log_n, width)
    ld = out_ld or width
    out_a = ctx.alloc(ld << (log_n + log_blowup))
    out_b = ctx.alloc(ld << (log_n + log_blowup))
    try:
        assert ctx.set_lde_fusion(True) is True           # on by default
        ctx.coset_lde(src, log_n, width, log_blowup, out=out_a, out_ld=ld)
        ctx.set_lde_fusion(False)
        ctx.coset_lde(src, log_n, width, log_blowup, out=out_b, out_ld=ld)
    finally:
        ctx.set_lde_fusion(True)
    a = out_a.download().reshape(-1, ld)[:, :width]
    b = out_b.download().reshape(-1, ld)[:, :width]
    assert (a == b).all()
    if width == 32:
        assert (a == oracle.coset_lde(src.download().reshape(-1, width), log_blowup, 31)).all()
    for x in (src, out_a, out_b):
        x.free()


def test_coset_lde_strided_output(ctx, oracle):
    # out_ld > width, as the prover uses for the two quotient chunks
    log_n = 9
    m = oracle.fill_uniform(SEED + 5, log_n, 4)
    out = ctx.alloc(8 << (log_n + 1))
    out.upload(np.zeros(8 << (log_n + 1), dtype=np.uint32))
    ctx.coset_lde(ctx.from_numpy(m), log_n, 4, out=out, out_ld=8, out_col=4)
    got = out.download().reshape(-1, 8)
    assert (got[:, 4:] == oracle.coset_lde(m, 1, 31)).all()
    assert (got[:, :4] == 0).all()


def test_poseidon2_known_answers(ctx, oracle):
    st = np.stack([np.arange(16), np.zeros(16)] + [np.random.default_rng(i).integers(0, P, 16) for i in range(62)]).astype(np.uint32)
    buf = ctx.from_numpy(st)
    ctx.poseidon2_permute(buf)
    got = buf.download().reshape(-1, 16)
    assert got[0].tolist() == KAT["poseidon2_iota"]
    assert got[1].tolist() == KAT["poseidon2_zero"]
    for i in range(2, 64):
        assert (got[i] == oracle.poseidon2(st[i])).all()


def test_poseidon2_extreme_states(ctx, oracle):
    """The device permutation runs on signed, unreduced representatives (poseidon2.cuh): drive it with the values that sit on
    the edges of every range argument -- 0, 1, P-1, (P-1)/2, (P+1)/2, powers of two -- alone and mixed, plus 4096 random states."""
    edge = np.array([0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 1 << 27, (1 << 27) - 1, 1 << 30, 0x0ffffffe, 0x78000000], dtype=np.uint32)
    rng = np.random.default_rng(2024)
    rows = [np.full(16, v, dtype=np.uint32) for v in edge]
    rows += [np.where(np.arange(16) % 2 == 0, a, b).astype(np.uint32) for a in edge[:6] for b in edge[:6]]
    rows += [rng.choice(edge, 16) for _ in range(512)]
    rows += [rng.integers(0, P, 16, dtype=np.uint32) for _ in range(4096)]
    st = np.stack(rows).astype(np.uint32)
    buf = ctx.from_numpy(st)
    ctx.poseidon2_permute(buf)
    got = buf.download().reshape(-1, 16)
    assert (got < P).all()
    for i in range(st.shape[0]):
        assert (got[i] == oracle.poseidon2(st[i])).all(), i


@pytest.mark.parametrize("fill", [0, 1, P - 1, (P + 1) // 2])
def test_hash_rows_and_p24_commit_of_constant_matrices(ctx, oracle, fill):
    m = np.full((256, 40), fill, dtype=np.uint32)
    got = ctx.hash_rows([(ctx.from_numpy(m), 40)], 256).download().reshape(-1, 8)
    assert (got == oracle.hash_rows([m])).all()
    cm = np.ascontiguousarray(m[:, :24].T)                      # column-major [cols][rows]
    got24 = ctx.merkle_commit_p24_colmajor(ctx.from_numpy(cm), 24, 8).download().reshape(-1, 8)
    assert (got24 == oracle.merkle_tree_p24_colmajor(cm)).all()


@pytest.mark.parametrize("height,widths", [(64, [8]), (300, [5]), (1024, [16]), (1024, [20]), (257, [1]), (512, [3, 9]), (128, [8, 8, 4, 1])])
def test_hash_rows_matches_oracle(ctx, oracle, height, widths):
    rng = np.random.default_rng(height)
    mats = [rng.integers(0, P, size=(height, w), dtype=np.uint32) for w in widths]
    dm = [(ctx.from_numpy(m), m.shape[1]) for m in mats]
    got = ctx.hash_rows(dm, height).download().reshape(-1, 8)
    assert (got == oracle.hash_rows(mats)).all()


@pytest.mark.parametrize("log_h,widths", [(0, [8]), (1, [8]), (5, [4]), (11, [16]), (12, [8]), (14, [24]), (13, [4, 4])])
def test_merkle_commit_matches_oracle(ctx, oracle, log_h, widths):
    rng = np.random.default_rng(log_h)
    mats = [rng.integers(0, P, size=(1 << log_h, w), dtype=np.uint32) for w in widths]
    dm = [(ctx.from_numpy(m), m.shape[1]) for m in mats]
    got = ctx.merkle_commit(dm, log_h).download().reshape(-1, 8)
    assert (got == oracle.merkle_tree(mats)).all()


def test_ntt_pass_hook_composes_to_bitrev_dft(ctx, oracle):
    # the two launches the benchmark times are exactly the forward transform
    log_n, width = 12, 16
    m = oracle.fill_uniform(SEED + 9, log_n, width)
    buf = ctx.from_numpy(m)
    ctx.ntt_pass(buf, buf, log_n, width, 0)
    ctx.ntt_pass(buf, buf, log_n, width, 1)
    got = buf.download().reshape(-1, width)
    assert (got[bitrev_perm(log_n)] == oracle.ntt(m)).all()


@pytest.mark.parametrize("shapes", [[(3, 3), (2, 2)], [(10, 16), (10, 5), (7, 8), (3, 4)], [(13, 24), (12, 8), (12, 3), (0, 6)],
                                    [(15, 8), (9, 40)]])
def test_merkle_commit_mixed_heights_matches_oracle(ctx, oracle, shapes):
    # an SP1 shard commits one matrix per chip, of different heights (p3-merkle-tree injection rule)
    rng = np.random.default_rng(len(shapes))
    mats = [rng.integers(0, P, size=(1 << lh, w), dtype=np.uint32) for lh, w in shapes]
    dm = [(ctx.from_numpy(m), m.shape[1], lh) for m, (lh, _) in zip(mats, shapes)]
    got = ctx.merkle_commit_mixed(dm).download().reshape(-1, 8)
    assert (got == oracle.merkle_tree_mixed(mats)).all()


@pytest.mark.parametrize("cols,log_rows", [(1, 0), (16, 3), (19, 6), (40, 10), (7, 13)])
def test_merkle_commit_p24_colmajor_matches_oracle(ctx, oracle, cols, log_rows):
    # RISC Zero layout: column-major polynomials, Poseidon2 width 24
    rng = np.random.default_rng(cols)
    mat = rng.integers(0, P, size=(cols, 1 << log_rows), dtype=np.uint32)
    got = ctx.merkle_commit_p24_colmajor(ctx.from_numpy(mat), cols, log_rows).download().reshape(-1, 8)
    assert (got == oracle.merkle_tree_p24_colmajor(mat)).all()


@pytest.mark.parametrize("count,log_size,log_blowup", [(1, 5, 2), (3, 8, 2), (40, 10, 2), (17, 12, 1), (64, 14, 2), (1, 20, 0), (3, 20, 2), (6, 20, 1)])
def test_colmajor_interpolate_and_expand_match_oracle(ctx, oracle, count, log_size, log_blowup):
    # RISC Zero Hal layout: `count` contiguous polynomials; 2^20-point polynomials take the NATIVE
    # contiguous-vector passes (ntt_colpass_kernel: transposes folded into the tiles' loads / stores), the rest the transposing adapter
    rng = np.random.default_rng(count)
    n = 1 << log_size
    evals_nat = rng.integers(0, P, size=(n, count), dtype=np.uint32)          # row-major view for the oracle
    coeffs = oracle.ntt(evals_nat, inverse=True)
    evals_br_colmajor = np.ascontiguousarray(evals_nat[bitrev_perm(log_size)].T)   # [count][n], bit-reversed order
    got_c = ctx.batch_interpolate_colmajor(ctx.from_numpy(evals_br_colmajor), count, log_size).download().reshape(count, n)
    assert (got_c == coeffs.T).all()
    # expansion: evaluations on 31 * <w_{n 2^b}>, bit-reversed = the oracle's coset LDE of the evaluations
    exp = oracle.coset_lde(evals_nat, log_blowup, 31)                         # [n << b][count]
    got_e = ctx.batch_expand_colmajor(ctx.from_numpy(np.ascontiguousarray(coeffs.T)), count, log_size, log_blowup, 31)
    assert (got_e.download().reshape(count, n << log_blowup) == exp.T).all()


@pytest.mark.parametrize("log_n,width,log_blowup,hw", [(6, 4, 1, 16), (10, 24, 1, 16), (12, 16, 2, 24), (9, 8, 0, 16), (11, 12, 3, 24)])
def test_commit_in_one_call_matches_oracle(ctx, oracle, log_n, width, log_blowup, hw):
    # the PCS commit of the boundary: LDE + tree + root
    m = oracle.fill_uniform(SEED + 3 * log_n, log_n, width)
    lde, tree, root = ctx.commit(ctx.from_numpy(m), log_n, width, log_blowup, hw)
    exp = oracle.coset_lde(m, log_blowup, 31)
    assert (lde.download().reshape(-1, width) == exp).all()
    otree = oracle.merkle_tree_hw(exp, hw)
    assert (tree.download().reshape(-1, 8) == otree).all()
    assert root.tolist() == otree[-1].tolist()
