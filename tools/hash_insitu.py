#!/usr/bin/env python3
"""the leaf-hash kernel alone against the same kernel inside a proof (one shard in flight).
Run under `rocprofv3 --kernel-trace` (durations per dispatch) and under `rocprofv3 --pmc <counter>` (one counter per pass):
    phase A: 3 warm + 6 proofs of the 2^20 x 256 headline shard, one in flight  -> its zk::hash_rows_vec_kernel dispatches are IN SITU
    phase B: 300 ms of sleep, then 3 warm + 6 isolated hash_rows launches over a 2^21 x 256 matrix (what the trace commitment hashes)
tools/hash_insitu_report.py turns the CSVs into a table (per dispatch, in dispatch order; the grid tells the two uses apart)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zktls_amd._lib import Params  # noqa: E402
from zktls_amd.device import Context  # noqa: E402

ctx = Context(0)
log_n, w = 20, 256
prm = Params(1, 100, 16)
trs = [ctx.gen_trace(1, s, log_n, w) for s in range(3)]
for i in range(9):
    ctx.prove_shard(trs[i % 3], log_n, w, [1, i], prm)
ctx.sync()
time.sleep(0.3)
lde = ctx.fill_uniform(2, log_n + 1, w)
dig = ctx.alloc(8 << (log_n + 1))
for _ in range(9):
    ctx.hash_rows([(lde, w)], 2 << log_n, out=dig)
ctx.sync()
ctx.close()
