"""exploration: the FIXED part of the term-parallel quotient kernel at the width of the SHA-256 chip -- a 608-column program with a
handful of terms (staging of 16 rows per group of 8 points + the reduction, no term work): python tools/airq_fixed.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O
from zktls_amd.device import Context, sha256_air, p2chip_air

ctx = Context(0)
V = O.air_var
for name, width, log_n in (("608 columns", 608, 20), ("360 columns", 360, 20), ("128 columns", 128, 20)):
    few = O.air_program(width, 1, [(O.SEL_ALL, [(1, [V(0), V(1)]), (O.P - 1, [V(2)])]), (O.SEL_TRANSITION, [(1, [V(3, True)]), (O.P - 1, [V(3)])])] * 2)
    tr = ctx.gen_trace(1, 0, log_n, width)
    lde = ctx.coset_lde(tr, log_n, width)
    progs = [("8 terms", few, [1])]
    if width == 608:
        progs.append(("SHA-256 chip", sha256_air(), list(range(16))))
    if width == 360:
        progs.append(("Poseidon2 chip", p2chip_air(), list(range(9))))
    for pname, prog, pub in progs:
        out = ctx.quotient_values_air(prog, lde, log_n, width, pub, [1, 2, 3, 4])
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.quotient_values_air(prog, lde, log_n, width, pub, [1, 2, 3, 4], out=out)
        ctx.sync()
        print("%-12s %-15s %.3f ms per quotient of 2^%d points" % (name, pname, (time.perf_counter() - t0) / 5 * 1e3, log_n + 1))
        out.free()
    tr.free(); lde.free()
