"""Full-size (BASELINE.json configs[1]: 2^20 rows x 256 columns) checks through
size-independent properties -- the oracle cannot produce 2^28-cell references in seconds:
  * inverse(forward(x)) == x over the whole matrix,
  * the committed LDE restricted to the original domain reproduces the trace (shift 1),
  * linearity of the LDE on a checksum of columns,
  * a complete shard proof that both the product's host verifier and the CPU oracle's
    verifier accept, with Merkle openings of the 2^21-leaf trees checked inside it."""
import numpy as np
import pytest

from zktls_amd._lib import Params
from zktls_amd.device import verify_shard

pytestmark = pytest.mark.gpu
P = 2013265921
SEED = 0x5A4B544C53
LOG_N, WIDTH = 20, 256


def test_fullsize_dft_round_trip(ctx):
    src = ctx.fill_uniform(SEED + 77, LOG_N, WIDTH)
    fwd = ctx.dft(src, LOG_N, WIDTH)
    back = ctx.dft(fwd, LOG_N, WIDTH, inverse=True)
    a = src.download_monty()
    b = back.download_monty()
    assert (a == b).all()
    assert not (fwd.download_monty(1 << 20) == a[: 1 << 20]).all()
    for x in (fwd, back, src):
        x.free()


def test_fullsize_lde_contains_trace_and_is_linear(ctx, oracle):
    width = 64                                   # keeps the host copies small; rows stay 2^20
    a = ctx.fill_uniform(SEED + 1, LOG_N, width)
    b = ctx.fill_uniform(SEED + 2, LOG_N, width)
    la = ctx.coset_lde(a, LOG_N, width, 1, 1)   # shift 1: even evaluation indices are the trace itself
    ha = a.download_monty().reshape(-1, width)
    hla = la.download_monty().reshape(-1, width)
    # evaluation index 2k sits at bit-reversed row bitrev_21(2k) = bitrev_20(k) (top half)
    idx = np.arange(1 << LOG_N, dtype=np.uint32)
    rev = np.zeros_like(idx)
    for bit in range(LOG_N):
        rev |= ((idx >> bit) & 1) << (LOG_N - 1 - bit)
    assert (hla[rev] == ha).all()
    # linearity on a column checksum: LDE(a + b) = LDE(a) + LDE(b) (canonical sums mod p)
    hb = b.download().reshape(-1, width)
    s = ((oracle.from_monty(ha).astype(np.uint64) + hb) % P).astype(np.uint32)
    ls = ctx.coset_lde(ctx.from_numpy(s), LOG_N, width, 1, 1).download().reshape(-1, width)
    lb = ctx.coset_lde(b, LOG_N, width, 1, 1).download().reshape(-1, width)
    assert (ls == ((oracle.from_monty(hla).astype(np.uint64) + lb) % P).astype(np.uint32)).all()


def test_fullsize_shard_proof_verifies(ctx, oracle):
    trace = ctx.gen_trace(SEED, 5, LOG_N, WIDTH)
    prm = Params(1, 100, 16)
    proof = ctx.prove_shard(trace, LOG_N, WIDTH, [1, 2, 3], prm)
    assert proof.size == 4 * (8 + 16 + 8 * WIDTH + 32 + 8 * LOG_N + 5 + 100 * (WIDTH + 8 + 16 * 21 + sum(4 + 8 * (20 - l) for l in range(20))))
    assert verify_shard(proof, LOG_N, WIDTH, [1, 2, 3], prm) == (0, 0)
    assert oracle.verify_shard(proof, LOG_N, WIDTH, [1, 2, 3], oracle.default_params(1, 100, 16)) == 0
    bad = proof.copy().view(np.uint32)
    bad[-3] = (int(bad[-3]) + 1) % P
    assert verify_shard(bad.view(np.uint8), LOG_N, WIDTH, [1, 2, 3], prm)[0] == -6
    # proving the same shard again gives the same bytes (deterministic PoW, no races)
    again = ctx.prove_shard(trace, LOG_N, WIDTH, [1, 2, 3], prm)
    assert again.tobytes() == proof.tobytes()


def test_fullsize_segment_proof_risc0_shape_verifies(ctx, oracle):
    # BASELINE.json configs[4]: one 2^20-cycle segment, RISC Zero's shape (blowup 4, fold 16, final 256
    # coefficients, 50 queries, Poseidon2 width 24); 128 columns keep the 4x LDE at 2 GiB
    from zktls_amd._lib import segment_params
    width = 128
    trace = ctx.gen_trace(SEED, 6, LOG_N, width)
    prm = segment_params()
    proof = ctx.prove_shard(trace, LOG_N, width, [1, 2, 3], prm)
    assert verify_shard(proof, LOG_N, width, [1, 2, 3], prm) == (0, 0)
    assert oracle.verify_shard(proof, LOG_N, width, [1, 2, 3], oracle.segment_params()) == 0
    bad = proof.copy().view(np.uint32)
    bad[-3] = (int(bad[-3]) + 1) % P
    assert verify_shard(bad.view(np.uint8), LOG_N, width, [1, 2, 3], prm)[0] == -6
    trace.free()


def test_fullsize_multi_chip_shard_verifies(ctx, oracle):
    # an SP1-like shard: a 2^20-row CPU-style table next to shorter, wider and narrower ones (219 M cells, one proof)
    from zktls_amd.device import verify_chips
    chips = [(20, 96), (20, 32), (19, 64), (18, 128), (16, 256), (14, 40)]
    dev = [(ctx.gen_trace(SEED, 10 + i, ln, w), ln, w) for i, (ln, w) in enumerate(chips)]
    prm = Params(1, 100, 16)
    proof = ctx.prove_chips(dev, [1, 2, 3], prm)
    lns, ws = [c[0] for c in chips], [c[1] for c in chips]
    assert verify_chips(proof, lns, ws, [1, 2, 3], prm) == (0, 0)
    assert oracle.verify_chips(proof, lns, ws, [1, 2, 3], oracle.default_params(1, 100, 16)) == 0
    bad = proof.copy().view(np.uint32)
    bad[-5] = (int(bad[-5]) + 1) % P
    assert verify_chips(bad.view(np.uint8), lns, ws, [1, 2, 3], prm)[0] == -6
    again = ctx.prove_chips(dev, [1, 2, 3], prm)
    assert again.tobytes() == proof.tobytes()
    for b, _, _ in dev:
        b.free()
