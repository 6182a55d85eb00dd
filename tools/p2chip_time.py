"""The Poseidon2 chip at size: 2^k Merkle openings of depth 16 of a random tree proven in-circuit (GPU box): python tools/p2chip_time.py [log_paths=16]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from zktls_amd._lib import Params
from zktls_amd.device import Context, verify_merkle_paths

log_paths = int(sys.argv[1]) if len(sys.argv) > 1 else 16
depth, SEED = 16, 0x5A4B544C53
ctx = Context(0)
d = ctx.fill_uniform(SEED, depth - 1, 16)
lde = ctx.coset_lde(d, depth - 1, 16)
tree = ctx.merkle_commit([(lde, 16)], depth).download().reshape(-1, 8)
levels, off = [], 0
for l in range(depth + 1):
    levels.append(tree[off:off + (1 << (depth - l))])
    off += 1 << (depth - l)
root = levels[-1][0]
prm = Params(1, 100, 16)
for lp in sorted({10, 14, log_paths}):
    n = 1 << lp
    idx = np.random.default_rng(lp).integers(0, 1 << depth, n).astype(np.uint32)
    leaves = levels[0][idx]
    sibs = np.ascontiguousarray(np.stack([levels[l][(idx >> l) ^ 1] for l in range(depth)], axis=1))
    t0 = time.perf_counter()
    tr, roots, log_n = ctx.p2chip_gen_merkle_trace(leaves, sibs, idx)
    ctx.sync()
    tg = time.perf_counter() - t0
    tr.free()
    ctx.prove_merkle_paths(leaves, sibs, idx, root, prm)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        proof = ctx.prove_merkle_paths(leaves, sibs, idx, root, prm)
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    assert verify_merkle_paths(proof, root, n, prm) == (0, 0)
    tv = time.perf_counter() - t0
    rows = 1 << log_n
    print("2^%d openings of depth %d = 2^%d rows x 360: trace (H2D of paths + kernel) %.1f ms, trace + proof %.1f ms (%.1f M permutations/s proven, %.2f G cells/s), proof %d bytes, verified in %.1f ms"
          % (lp, depth, log_n, tg * 1e3, dt * 1e3, n * depth / dt / 1e6, rows * 360 / dt / 1e9, proof.size, tv * 1e3))
