"""Latency of small proofs (one shard in flight): python tools/latency_small.py [log_n ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zktls_amd._lib import Params
from zktls_amd.device import Context

ctx = Context(0)
prm = Params(1, 100, 16)
width = int(os.environ.get("WIDTH", "16"))
for log_n in ([int(x) for x in sys.argv[1:]] or (6, 8, 10, 12, 14, 16)):
    trace = ctx.gen_trace(1, 0, log_n, width)
    for _ in range(3):
        ctx.prove_shard(trace, log_n, width, [1, 2, 3], prm)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.prove_shard(trace, log_n, width, [1, 2, 3], prm)
    dt = (time.perf_counter() - t0) / reps
    print("2^%d x %d: %.2f ms per proof" % (log_n, width, dt * 1e3))
    trace.free()
