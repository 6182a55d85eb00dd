#!/usr/bin/env python3
"""Minimal driver for profiler runs (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes): one shard proof, then the four
launches of its trace LDE (zkhip_ntt_pass which = 2..5: inverse strided / inverse contiguous / forward strided-out / forward
contiguous) a few times each on the proving context's own workspaces -- the kernels and the buffer placement bench.py's
`roofline` section times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zktls_amd._lib import Params  # noqa: E402
from zktls_amd.device import Context  # noqa: E402

ctx = Context(0)
log_n, w = 20, 256
trace = ctx.gen_trace(0x5A4B544C53, 0, log_n, w)
ctx.prove_shard(trace, log_n, w, [1, 2, 3], Params(1, 100, 16))
for _ in range(4):
    for which in (2, 3, 4, 5):
        ctx.ntt_pass(trace, None, log_n, w, which)
ctx.sync()
ctx.close()
