"""Quotient of narrow constraint programs: time zkhip_quotient_values_air per launch (GPU box): python tools/airq_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import airs
from zktls_amd.device import Context, air_synthetic

ctx = Context(0)
log_n = 18
for name, width, prog, pub in [("synthetic 4", 4, air_synthetic(4, 3), [1, 2, 3]), ("synthetic 8", 8, air_synthetic(8, 3), [1, 2, 3]),
                               ("synthetic 32", 32, air_synthetic(32, 3), [1, 2, 3]), ("synthetic 128", 128, air_synthetic(128, 3), [1, 2, 3]),
                               ("counter 8", 8, airs.counter_program(8), [3, 5]), ("counter 64", 64, airs.counter_program(64), [3, 5]),
                               ("fibonacci 4", 4, airs.fibonacci_program(), [0, 1, 2])]:
    tr = ctx.gen_trace(1, 0, log_n, width)
    lde = ctx.coset_lde(tr, log_n, width)
    out = ctx.quotient_values_air(prog, lde, log_n, width, pub, [1, 2, 3, 4])
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.quotient_values_air(prog, lde, log_n, width, pub, [1, 2, 3, 4], out=out)
    ctx.sync()
    print("%-14s %5d words: %.3f ms per quotient of 2^%d points" % (name, prog.size, (time.perf_counter() - t0) / 10 * 1e3, log_n + 1))
    tr.free(); lde.free(); out.free()
