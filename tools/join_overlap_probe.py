#!/usr/bin/env python3
"""the tree's four joins one after the other against k of them in flight (own context = stream, own key, own host thread each):
how much of a join's 31 ms is the device waiting for the host?  usage: python tools/join_overlap_probe.py [joins=4] [proofs per join=16]"""
import os
import sys
import threading
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zktls_amd._lib import Params  # noqa: E402
from zktls_amd.device import Context, set_lockstep  # noqa: E402

nt, nj = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 16
log_n, width, q, pb = 20, 256, 100, 16
prm = Params(1, q, pb)
set_lockstep(16, 6)
ctxs = [Context(0) for _ in range(nt)]
tr = ctxs[0].gen_trace(1, 0, log_n, width)
pv = [[1, 2, 3, 4, 5, 6, 7, 8, s] for s in range(nj * nt)]
shards = [ctxs[0].prove_shard(tr, log_n, width, pv[s], prm) for s in range(nj * nt)]
keys = [c.shard_verifier_setup(log_n, width, q, pb, 9, prm, n_proofs=nj) for c in ctxs]
assert all(k.root.tolist() == keys[0].root.tolist() for k in keys)


def join(w, j, out):
    out[j] = ctxs[w].prove_shard_verifier(keys[w], shards[nj * j:nj * (j + 1)], log_n, width, pv[nj * j:nj * (j + 1)], prm, prm)


ref = [None] * nt
for j in range(nt):
    join(0, j, ref)
for k in (1, 2, 4):
    if k > nt:
        break
    best = 1e9
    for rep in range(4):
        out = [None] * nt
        t0 = time.perf_counter()
        if k == 1:
            for j in range(nt):
                join(0, j, out)
        else:
            def worker(w):
                for j in range(w, nt, k):
                    join(w, j, out)
            th = [threading.Thread(target=worker, args=(w,)) for w in range(k)]
            [t.start() for t in th]
            [t.join() for t in th]
        best = min(best, time.perf_counter() - t0)
        assert all(bytes(a) == bytes(b) for a, b in zip(out, ref))
    print("%d joins of %d, %d in flight: %.1f ms" % (nt, nj, k, best * 1e3), flush=True)

# the same through the library's own entry (pooled contexts that keep the shape's key, workers on native threads)
from zktls_amd.device import prove_shard_verifier_batch  # noqa: E402
for k in (1, 2, 4):
    if k > nt:
        break
    prove_shard_verifier_batch(shards, nj, log_n, width, pv, prm, prm, devices=[0], in_flight=k)
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        out, vk = prove_shard_verifier_batch(shards, nj, log_n, width, pv, prm, prm, devices=[0], in_flight=k)
        best = min(best, time.perf_counter() - t0)
        assert all(bytes(a) == bytes(b) for a, b in zip(out, ref)) and vk.tolist() == keys[0].root.tolist()
    print("zkhip_prove_shard_verifier_batch, %d joins of %d, %d in flight: %.1f ms" % (nt, nj, k, best * 1e3), flush=True)
