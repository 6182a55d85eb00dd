#!/usr/bin/env python3
"""the bench line's `multichip.compressed` shard -- the six chips as ONE keyed machine (version 11: constraint programs, interaction tables, a cross-table bus,
32 preprocessed columns) -- one at a time: the phases of zkhip_prove_machine_keyed (A/B build, ZKHIP_CHIPS_TIMING), beside tools/multichip_breakdown.py's unkeyed form"""
import os
import sys
import time
os.environ["ZKHIP_CHIPS_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _ab  # noqa: E402,F401
from zktls_amd._lib import Params  # noqa: E402
from zktls_amd.device import Context, Sp1ShapedShard  # noqa: E402

ctx = Context(0)
prm = Params(1, 100, 16)
shape = Sp1ShapedShard()
key, keep = shape.setup(ctx, 0x5A4B544C53, prm)
traces = [shape.gen_traces(ctx, 0x5A4B544C53, s) for s in range(2)]
ctx.sync()
for rep in range(4):
    t0 = time.perf_counter()
    pf = ctx.prove_machine_keyed(key, shape.main_chips(traces[rep % 2]), shape.programs, shape.tables, [1, 2, 3, 4, 5, 6, 7, 8, rep], prm)
    print("keyed multichip shard: %.2f ms, %d bytes" % ((time.perf_counter() - t0) * 1e3, pf.size), flush=True)
key.close()
ctx.close()
