#!/usr/bin/env python3
"""64 transcript proofs -> ONE proof two ways: the flat air-mode join of 64 (what bench.py's batch64.compressed times) against a tree -- joins of J
in flight (own context, key and host thread each), then machine mode over the joins.  usage: python tools/compress64_tree_probe.py [J=16]"""
import hashlib
import os
import sys
import threading
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zktls_amd._lib import Params  # noqa: E402
from zktls_amd.device import (Context, InnerMachine, prove_transcripts, set_lockstep, sha256_air, sha256_padding_publics, shard_verifier_describe,  # noqa: E402
                              verify_machine_recursive, verify_shard_recursive)

J = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n, nbytes, q, pb, log_n, W = 64, 13221, 100, 16, 14, 640
nt = n // J
prm = Params(1, q, pb)
set_lockstep(16, 6)
prog = sha256_air()
msgs = [bytes((7 * i + 3 * p + 1) & 0xff for i in range(nbytes)) for p in range(n)]
inner, pubs = [], []
for m, (d, pf) in zip(msgs, prove_transcripts(msgs, prm, devices=[0], keyed=False)[1]):
    assert d == hashlib.sha256(m).digest()
    limbs = []
    for k in range(8):
        w = int.from_bytes(d[4 * k:4 * k + 4], "big")
        limbs += [w & 0xffff, w >> 16]
    inner.append(pf), pubs.append(limbs + [int(x) for x in sha256_padding_publics(len(m))])
npub = len(pubs[0])
ctxs = [Context(0) for _ in range(nt)]
flat_key = ctxs[0].shard_verifier_setup(log_n, W, q, pb, npub, prm, n_proofs=n, program=prog)
ctxs[0].prove_shard_verifier(flat_key, inner, log_n, W, pubs, prm, prm, program=prog)
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    flat = ctxs[0].prove_shard_verifier(flat_key, inner, log_n, W, pubs, prm, prm, program=prog)
    best = min(best, time.perf_counter() - t0)
print("flat join of %d: %.1f ms, %d bytes" % (n, best * 1e3, flat.size), flush=True)
flat_key.close()

keys = [c.shard_verifier_setup(log_n, W, q, pb, npub, prm, n_proofs=J, program=prog) for c in ctxs]
chips = []
for i in range(9):
    p, ln, mw, pw = shard_verifier_describe(log_n, W, q, pb, npub, i, 0, J, program=prog)
    t, _, _, _ = shard_verifier_describe(log_n, W, q, pb, npub, i, 1, J, program=prog)
    chips.append(dict(ln=ln, W=mw, Pw=pw, prog=p, tab=t))
im = InnerMachine(chips, keys[0].root, q, pb, J * npub)
tkey = ctxs[0].machine_verifier_setup(im, prm, nt)
jp = [[v for p in pubs[J * j:J * (j + 1)] for v in p] for j in range(nt)]


def tree(k):
    out = [None] * nt

    def worker(w):
        for j in range(w, nt, k):
            out[j] = ctxs[w].prove_shard_verifier(keys[w], inner[J * j:J * (j + 1)], log_n, W, pubs[J * j:J * (j + 1)], prm, prm, program=prog)
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(w,)) for w in range(k)]
    [t.start() for t in th]
    [t.join() for t in th]
    t1 = time.perf_counter()
    top = ctxs[0].prove_machine_verifier(tkey, im, out, jp, prm)
    return t1 - t0, time.perf_counter() - t1, out, top


for k in (1, 2, nt):
    tree(k)
    best = (1e9, 0, 0)
    for rep in range(3):
        tj, tt, out, top = tree(k)
        if tj + tt < best[0]:
            best = (tj + tt, tj, tt)
    print("tree: %d joins of %d, %d in flight: %.1f ms = joins %.1f + top %.1f; %d bytes" % (nt, J, k, best[0] * 1e3, best[1] * 1e3, best[2] * 1e3, top.size), flush=True)
for j in range(nt):
    assert verify_shard_recursive(out[j], log_n, W, q, pb, jp[j], keys[0].root, prm, n_proofs=J, program=prog) == (0, 0)
assert verify_machine_recursive(im, top, [v for p in jp for v in p], tkey.root, prm, nt) == (0, 0)
print("verified", flush=True)
