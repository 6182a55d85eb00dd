#!/usr/bin/env python3
"""where the tree's time goes (A/B build: ZKHIP_REC_TIMING prints the phases of zkhip_prove_shard_verifier and zkhip_prove_machine_verifier to stderr):
64 headline shard proofs -> 4 joins of 16 -> one proof"""
import os
import sys
import time
os.environ["ZKHIP_REC_TIMING"] = "1"
os.environ["ZKHIP_CHIPS_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _ab  # noqa: E402,F401
from zktls_amd._lib import Params  # noqa: E402
from zktls_amd.device import Context, InnerMachine, shard_verifier_describe, verify_machine_recursive  # noqa: E402

ctx = Context(0)
log_n, width, q, pb, nj, nt = 20, 256, 100, 16, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 4
prm = Params(1, q, pb)
tr = ctx.gen_trace(1, 0, log_n, width)
pv = [[1, 2, 3, 4, 5, 6, 7, 8, s] for s in range(nj * nt)]
shards = [ctx.prove_shard(tr, log_n, width, pv[s], prm) for s in range(nj * nt)]
jkey = ctx.shard_verifier_setup(log_n, width, q, pb, 9, prm, n_proofs=nj)
joins = [ctx.prove_shard_verifier(jkey, shards[nj * j:nj * (j + 1)], log_n, width, pv[nj * j:nj * (j + 1)], prm, prm) for j in range(nt)]
jp = [[v for p in pv[nj * j:nj * (j + 1)] for v in p] for j in range(nt)]
chips = []
for i in range(8):
    p, ln, mw, pw = shard_verifier_describe(log_n, width, q, pb, 9, i, 0, nj)
    t, _, _, _ = shard_verifier_describe(log_n, width, q, pb, 9, i, 1, nj)
    chips.append(dict(ln=ln, W=mw, Pw=pw, prog=p, tab=t))
im = InnerMachine(chips, jkey.root, q, pb, 9 * nj)
tkey = ctx.machine_verifier_setup(im, prm, nt)
for rep in range(3):
    t0 = time.perf_counter()
    top = ctx.prove_machine_verifier(tkey, im, joins, jp, prm)
    print("top over %d joins: %.1f ms, %d bytes" % (nt, (time.perf_counter() - t0) * 1e3, top.size), flush=True)
assert verify_machine_recursive(im, top, [v for p in jp for v in p], tkey.root, prm, nt) == (0, 0)
