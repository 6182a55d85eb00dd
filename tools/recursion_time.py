#!/usr/bin/env python3
"""the shard verifier machine at the headline shape: setup, prove (first / warm), verify; a HIP-event-free wall-clock view"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zktls_amd._lib import Params  # noqa: E402
from zktls_amd.device import Context, verify_shard_recursive  # noqa: E402

ctx = Context(0)
log_n, width, q, pb = 20, 256, 100, 16
pubs = list(range(1, 10))
iprm, prm = Params(1, q, pb), Params(1, 100, 16)
tr = ctx.gen_trace(1, 0, log_n, width)
inner = [ctx.prove_shard(tr, log_n, width, pubs[:-1] + [s], iprm) for s in range(4)]
t0 = time.perf_counter()
key = ctx.shard_verifier_setup(log_n, width, q, pb, len(pubs), prm)
ctx.sync()
print("setup %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for rep in range(6):
    t0 = time.perf_counter()
    outer = ctx.prove_shard_verifier(key, inner[rep % 4], log_n, width, pubs[:-1] + [rep % 4], iprm, prm)
    dt = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    rc = verify_shard_recursive(outer, log_n, width, q, pb, pubs[:-1] + [rep % 4], key.root, prm)
    print("prove %.1f ms, verify %.1f ms rc %s, inner %d B, outer %d B" % (dt, (time.perf_counter() - t0) * 1e3, rc, inner[0].size, outer.size))
key.close()
# the join: sixteen headline proofs -> one proof
n = 16
pv = [pubs[:-1] + [s] for s in range(n)]
tr16 = [ctx.prove_shard(tr, log_n, width, pv[s], iprm) for s in range(n)]
t0 = time.perf_counter()
key = ctx.shard_verifier_setup(log_n, width, q, pb, len(pubs), prm, n_proofs=n)
ctx.sync()
print("join setup %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for rep in range(4):
    t0 = time.perf_counter()
    outer = ctx.prove_shard_verifier(key, tr16, log_n, width, pv, iprm, prm)
    dt = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    rc = verify_shard_recursive(outer, log_n, width, q, pb, [v for p in pv for v in p], key.root, prm, n_proofs=n)
    print("join 16: prove %.1f ms (%.2f per inner proof), verify %.1f ms rc %s, inner %d B total, outer %d B" % (dt, dt / n, (time.perf_counter() - t0) * 1e3, rc, sum(x.size for x in tr16), outer.size))
key.close()
ctx.close()
