#!/usr/bin/env python3
"""Time the column-major (RISC Zero Hal layout) interpolate / expand-by-4 entries on `count` polynomials of 2^20 coefficients:
native contiguous-vector passes (ntt_colpass_kernel).  HIP events on the context's stream."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from zktls_amd.device import Context  # noqa: E402

count, log_size = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 20
st = torch.cuda.Stream()
ctx = Context(0, stream=st.cuda_stream)
n = 1 << log_size
with torch.cuda.stream(st):
    ev = torch.empty(count * n, dtype=torch.int32, device="cuda")
    co = torch.empty(count * n, dtype=torch.int32, device="cuda")
    ex = torch.empty(4 * count * n, dtype=torch.int32, device="cuda")
b_ev, b_co, b_ex = ctx.wrap(ev), ctx.wrap(co), ctx.wrap(ex)
ctx.fill_uniform(1, log_size, count, out=b_ev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn, elems_moved in (("interpolate", lambda: ctx.batch_interpolate_colmajor(b_ev, count, log_size, out=b_co), 2 * count * n),
                              ("expand x4", lambda: ctx.batch_expand_colmajor(b_co, count, log_size, 2, 31, out=b_ex), 4 * 2 * count * n)):
    for _ in range(3):
        fn()
    e0.record(st)
    for _ in range(10):
        fn()
    e1.record(st)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-12s %d x 2^%d: %.3f ms  (%.2f TB/s over the 8 B per element and pass it moves)" % (name, count, log_size, ms, elems_moved * 8 / ms / 1e9))
ctx.close()
