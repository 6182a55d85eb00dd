import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import _ab
from zktls_amd._lib import Params
from zktls_amd.device import Context
ctx = Context(0)
log_n, width = 20, 256
prm = Params(1, 100, 16)
tr = ctx.gen_trace(1, 0, log_n, width)
ts = []
for rep in range(12):
    t0 = time.perf_counter()
    pf = ctx.prove_shard(tr, log_n, width, [1, 2, 3, rep], prm)
    ts.append((time.perf_counter() - t0) * 1e3)
print("FRI_GRAPH=%s: min %.2f median %.2f ms" % (os.environ.get("ZKHIP_FRI_GRAPH"), min(ts[2:]), sorted(ts[2:])[5]))
