#!/usr/bin/env python3
"""one headline shard proof at a time (one in flight): run under `rocprofv3 --kernel-trace` and feed the trace to gap_report.py to see
where the latency of a single proof goes (kernel time against idle gaps between the host round trips)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zktls_amd._lib import Params  # noqa: E402
from zktls_amd.device import Context  # noqa: E402
ctx = Context(0)
log_n, width = 20, 256
prm = Params(1, 100, 16)
tr = ctx.gen_trace(1, 0, log_n, width)
for rep in range(6):
    t0 = time.perf_counter()
    pf = ctx.prove_shard(tr, log_n, width, [1, 2, 3, rep], prm)
    print("prove_shard %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
ctx.close()
