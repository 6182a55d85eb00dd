#!/usr/bin/env python3
"""Minimal driver for profiler runs: launches the NTT pass kernels (both passes of the
forward transform) a few times on a 2^20 x 256 matrix.  Used under rocprofv3 --pmc."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zktls_amd.device import Context  # noqa: E402

ctx = Context(0)
log_n, w = 20, 256
src = ctx.fill_uniform(1, log_n, w)
dst = ctx.alloc(w << log_n)
for _ in range(4):
    ctx.ntt_pass(src, dst, log_n, w, 0)
    ctx.ntt_pass(src, dst, log_n, w, 1)
ctx.sync()
ctx.close()
