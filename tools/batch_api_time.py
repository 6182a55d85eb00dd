"""exploration: zkhip_prove_shards (the C-ABI batch entry) on 8 shards of 2^20 x 256, wall time per shard for several in_flight values"""
import sys, time
sys.path.insert(0, "/root/repo")
from zktls_amd.device import Context, prove_shards
from zktls_amd._lib import Params
ctx = Context(0)
log_n, w = 20, 256
traces = [ctx.gen_trace(0x5A4B544C53, s, log_n, w) for s in range(8)]
ctx.sync()
pvs = [[1, 2, s] for s in range(8)]
for k in (1, 2, 4, 6):
    prove_shards(traces[:k], log_n, w, pvs[:k], Params(1, 100, 16), in_flight=k)        # warm caches / clocks
    t = time.time()
    prove_shards(traces, log_n, w, pvs, Params(1, 100, 16), in_flight=k)
    dt = time.time() - t
    print("in_flight %d: %.1f ms per shard (contexts cached from the warm-up call)" % (k, dt * 1e3 / 8))
