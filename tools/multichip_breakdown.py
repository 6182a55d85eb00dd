#!/usr/bin/env python3
"""the bench line's `multichip` shard (six chips with in-table LogUp pairs) one at a time: the phases of zkhip_prove_chips (A/B build,
ZKHIP_CHIPS_TIMING) -- run under `rocprofv3 --kernel-trace --stats` for its kernels"""
import os
import sys
import time
os.environ["ZKHIP_CHIPS_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _ab  # noqa: E402,F401
from zktls_amd._lib import Params  # noqa: E402
from zktls_amd.device import Context  # noqa: E402

ctx = Context(0)
spec = [(20, 96), (20, 32), (19, 64), (18, 128), (16, 256), (14, 40)]
pairs = [max(1, w // 32) for _, w in spec]
prm = Params(1, 100, 16)
bufs = [(ctx.gen_trace_logup(0x5A4B544C53, 7000 + j, ln, w, q), ln, w, q) for j, ((ln, w), q) in enumerate(zip(spec, pairs))]
ctx.sync()
for rep in range(4):
    t0 = time.perf_counter()
    pf = ctx.prove_chips(bufs, [1, 2, 3, rep], prm)
    print("multichip shard: %.2f ms, %d bytes" % ((time.perf_counter() - t0) * 1e3, pf.size), flush=True)
ctx.close()
