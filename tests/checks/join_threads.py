"""GPU: eight host threads, each with its own context, joining 1..4 shard proofs forty times: every repeat byte-equal to the first, the first
accepted by the host verifier (the join fills work lists on a host pool of its own and keeps a pinned block per context)"""
import sys, threading, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zktls_amd._lib import Params
from zktls_amd.device import Context, verify_shard_recursive
log_n, width, q, pb = 7, 16, 6, 2
iprm, prm = Params(1, q, pb), Params(1, 12, 4)
errs = []
def worker(tid):
    try:
        ctx = Context(0)
        n = 1 + tid % 4
        pv = [[tid, s, 7] for s in range(n)]
        inner = []
        for s in range(n):
            tr = ctx.gen_trace(99, 10 * tid + s, log_n, width)
            inner.append(ctx.prove_shard(tr, log_n, width, pv[s], iprm)); tr.free()
        key = ctx.shard_verifier_setup(log_n, width, q, pb, 3, prm, n_proofs=n)
        first = None
        for rep in range(40):
            outer = ctx.prove_shard_verifier(key, inner, log_n, width, pv, iprm, prm)
            if first is None:
                first = outer.tobytes()
                assert verify_shard_recursive(outer, log_n, width, q, pb, [v for p in pv for v in p], key.root, prm, n_proofs=n) == (0, 0)
            assert outer.tobytes() == first, (tid, rep)
        key.close(); ctx.close()
    except Exception as e:
        errs.append((tid, repr(e)))
ts = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
t0 = time.time()
[t.start() for t in ts]; [t.join() for t in ts]
print("8 threads x 40 joins in %.1f s, errors: %s" % (time.time() - t0, errs))
