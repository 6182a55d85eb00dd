"""One 2^20 x 256 shard in each of the preset FRI configurations of include/zkhip.h (GPU box): python tools/shape_time.py"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zktls_amd._lib import Params
from zktls_amd.device import Context, verify_shard
ctx = Context(0)
for name, prm in (("SP1 core", Params(1, 100, 16)), ("SP1 compress", Params(2, 50, 16)), ("RISC0", Params(2, 50, 0, 0, 4, 8, 24))):
    tr = ctx.gen_trace(1, 0, 20, 256)
    ctx.prove_shard(tr, 20, 256, [1], prm); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3): pf = ctx.prove_shard(tr, 20, 256, [1], prm)
    dt = (time.perf_counter() - t0) / 3
    assert verify_shard(pf, 20, 256, [1], prm) == (0, 0)
    print("%s: 2^20 x 256, %.1f ms per proof (one in flight), %d bytes" % (name, dt * 1e3, pf.size))
    tr.free()
