#!/usr/bin/env python3
"""Minimal driver for profiler runs of the fused-LDE launches: one LDE, then which = 6 (first inverse pass, block form), 7 (fused
middle launch) and 5 (second forward pass) a few times each on the context's own workspaces (zkhip_ntt_pass)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--ab" in sys.argv:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _ab  # noqa: F401
from zktls_amd.device import Context  # noqa: E402

ctx = Context(0)
log_n, w = 20, 256
trace = ctx.gen_trace(0x5A4B544C53, 0, log_n, w)
out = ctx.alloc((w << log_n) * 2)
ctx.coset_lde(trace, log_n, w, out=out)
for _ in range(4):
    for which in (6, 7, 5, 3):
        ctx.ntt_pass(trace, None, log_n, w, which)
ctx.sync()
ctx.close()
