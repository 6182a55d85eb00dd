#!/usr/bin/env python3
"""the join of sixteen headline shard proofs under different OUTER proof shapes: SP1 core (blowup 2, 100 queries, 16 PoW bits) and SP1 compress
(blowup 4, 50 queries, 16 PoW bits: the shape sp1-recursion's compress stage uses, [RECALLED]) -- time, bytes, host verification"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zktls_amd._lib import Params  # noqa: E402
from zktls_amd.device import Context, verify_shard_recursive  # noqa: E402

ctx = Context(0)
log_n, width, q, pb, n = 20, 256, 100, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 16
iprm = Params(1, q, pb)
pv = [[1, 2, 3, 4, 5, 6, 7, 8, s] for s in range(n)]
tr = ctx.gen_trace(1, 0, log_n, width)
inner = [ctx.prove_shard(tr, log_n, width, pv[s], iprm) for s in range(n)]
tr.free()
shapes = [("core shape (blowup 2, 100 queries)", Params(1, 100, 16)), ("compress shape (blowup 4, 50 queries)", Params(2, 50, 16)), ("blowup 8, 33 queries", Params(3, 33, 16))]
for name, prm in shapes[:3 if n <= 64 else (2 if n <= 68 else 1)]:      # (a 2^22-row Poseidon2 chip takes blowup 2 only: its 384-word rows)
    t0 = time.perf_counter()
    key = ctx.shard_verifier_setup(log_n, width, q, pb, 9, prm, n_proofs=n)
    ctx.sync()
    ts = (time.perf_counter() - t0) * 1e3
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        outer = ctx.prove_shard_verifier(key, inner, log_n, width, pv, iprm, prm)
        best = min(best, (time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    rc = verify_shard_recursive(outer, log_n, width, q, pb, [v for p in pv for v in p], key.root, prm, n_proofs=n)
    tv = (time.perf_counter() - t0) * 1e3
    print("%-40s setup %6.1f ms, join %6.1f ms, outer %8d B = 1 / %.1f of %d B, host verify %.2f ms rc %s" % (name, ts, best, outer.size, sum(x.size for x in inner) / outer.size, sum(x.size for x in inner), tv, rc), flush=True)
    key.close()
ctx.close()
