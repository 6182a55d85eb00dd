"""exploration: many proofs on concurrent contexts; every proof must verify and repeat byte for byte"""
import sys, threading, hashlib
sys.path.insert(0, "/root/repo")
from zktls_amd.device import Context, verify_shard, verify_chips
from zktls_amd._lib import Params
errors = []
def worker(w):
    try:
        ctx = Context(0)
        shapes = [(12, 32, Params(1, 20, 8)), (13, 64, Params(1, 30, 10)), (12, 16, Params(2, 20, 0, 0, 4, 4, 24)), (11, 24, Params(1, 20, 8, 2))]
        seen = {}
        for it in range(60):
            ln, w_, prm = shapes[(it + w) % len(shapes)]
            s = (it // 4) % 3
            tr = ctx.gen_trace_logup(7, s, ln, w_, prm.logup_pairs) if prm.logup_pairs else ctx.gen_trace(7, s, ln, w_)
            pf = ctx.prove_shard(tr, ln, w_, [s], prm)
            if verify_shard(pf, ln, w_, [s], prm) != (0, 0): errors.append(("verify", w, it))
            key = (ln, w_, s, prm.log_blowup, prm.logup_pairs)
            h = hashlib.sha256(pf.tobytes()).hexdigest()
            if seen.setdefault(key, h) != h: errors.append(("nondeterministic", w, it))
            tr.free()
            if it % 10 == 0:
                chips = [(ctx.gen_trace(9, 0, 12, 16), 12, 16), (ctx.gen_trace(9, 1, 9, 8), 9, 8), (ctx.gen_trace(9, 2, 9, 4), 9, 4)]
                cp = ctx.prove_chips(chips, [it], Params(1, 10, 4))
                if verify_chips(cp, [12, 9, 9], [16, 8, 4], [it], Params(1, 10, 4)) != (0, 0): errors.append(("chips", w, it))
                for b, _, _ in chips: b.free()
        ctx.close()
    except Exception as e:
        errors.append(("exception", w, repr(e)))
ts = [threading.Thread(target=worker, args=(w,)) for w in range(4)]
[t.start() for t in ts]; [t.join() for t in ts]
print("errors:", errors[:5], "count", len(errors))
