"""exploration: four host threads, each with its own context, proving shards of DIFFERENT random shapes at the same time (HIP-graph
re-capture, workspace growth and kernel launches interleave across threads); every proof must equal the oracle's bytes."""
import sys, threading, time
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np
import oracle_lib as O
from zktls_amd.device import Context, verify_shard
from zktls_amd._lib import Params
O.set_threads(2)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
errors, counts = [], [0] * 4
def worker(t):
    rng = np.random.default_rng(1000 + t)
    ctx = Context(0)
    t0 = time.time()
    try:
        while time.time() - t0 < budget:
            log_n = int(rng.integers(6, 14)); width = 4 * int(rng.integers(1, 20)); b = int(rng.integers(1, 3)); K = int(rng.integers(1, 4))
            fs = [f for f in range(0, min(log_n, 8) + 1) if (log_n - f) % K == 0]
            if not fs: continue
            F = int(rng.choice(fs)); hw = int(rng.choice([16, 24]))
            shape = (b, int(rng.integers(1, 12)), int(rng.integers(0, 6)), 0, K, F, hw)
            shard = int(rng.integers(0, 1000)); pub = [t, shard]
            d = ctx.gen_trace(77, shard, log_n, width)
            pf = ctx.prove_shard(d, log_n, width, pub, Params(*shape))
            exp = O.prove_shard(O.gen_trace(77, shard, log_n, width), pub, O.default_params(*shape))
            if pf.tobytes() != exp.tobytes(): errors.append((t, log_n, width, shape)); break
            if verify_shard(pf, log_n, width, pub, Params(*shape)) != (0, 0): errors.append(("verify", t, log_n, width, shape)); break
            d.free(); counts[t] += 1
    except Exception as e:
        errors.append((t, repr(e)))
    ctx.close()
ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
[t.start() for t in ths]; [t.join() for t in ths]
print("errors:", errors if errors else "none", "| proofs per thread:", counts)
