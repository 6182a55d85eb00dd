#!/usr/bin/env python3
"""The one-launch LDE of 2^11 .. 2^15 rows (csrc/ntt_small.hip) against the pass kernels it replaces at these heights (A/B build:
ZKHIP_LDE_SMALL=0 switches it off).  Operator level: ms per coset LDE at blowup 2 and the bytes per second of the 12 B per cell it must move;
then BASELINE configs[2] (64 transcripts, lock-step lanes) either way.  usage: python tools/small_lde_time.py [op|batch] (run once per setting)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.dirname(os.path.abspath(__file__))]
import _ab  # noqa: F401,E402
from zktls_amd._lib import Params  # noqa: E402
from zktls_amd.device import Context, prove_transcripts, set_lockstep  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "op"
print("ZKHIP_LDE_SMALL =", os.environ.get("ZKHIP_LDE_SMALL", "(unset: default gate)"), " ZKHIP_TILE_FIX =", os.environ.get("ZKHIP_TILE_FIX", "(unset: on)"), flush=True)
if what == "op":
    ctx = Context(0)
    for log_n in (11, 12, 13, 14, 15):
        for w in (640, 256, 64, 8):
            src = ctx.fill_uniform(3, log_n, w)
            out = ctx.alloc((w << log_n) * 2)
            for _ in range(3):
                ctx.coset_lde(src, log_n, w, out=out)
            ctx.sync()
            reps = 200
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.coset_lde(src, log_n, w, out=out)
            ctx.sync()
            dt = (time.perf_counter() - t0) / reps
            print("2^%d x %4d: %8.1f us per LDE, %7.1f GB/s of 12 B per cell" % (log_n, w, dt * 1e6, 12.0 * (w << log_n) / dt / 1e9), flush=True)
            src.free(); out.free()
    # many matrices at once, as a lock-step batch issues them: 64 x (2^14 x 640) in one stream, back to back
    log_n, w = 14, 640
    srcs = [ctx.fill_uniform(10 + i, log_n, w) for i in range(16)]
    out = ctx.alloc((w << log_n) * 2)
    for s in srcs:
        ctx.coset_lde(s, log_n, w, out=out)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(4):
        for s in srcs:
            ctx.coset_lde(s, log_n, w, out=out)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 64
    print("64 x (2^14 x 640) back to back: %.1f us each, %.1f GB/s of 12 B per cell" % (dt * 1e6, 12.0 * (w << log_n) / dt / 1e9))
else:
    prm = Params(1, 100, 16)
    base = open(os.path.join(ROOT, "tests", "golden", "reference", "guest_input0.cbor"), "rb").read()
    msgs = [base + i.to_bytes(4, "little") for i in range(64)]
    for batch, lanes in ((16, 6), (8, 8)):
        set_lockstep(batch, lanes)
        prove_transcripts(msgs, prm, devices=[0])
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            prove_transcripts(msgs, prm, devices=[0])
            best = min(best, time.perf_counter() - t0)
        print("64 transcripts, lock-step %d x %d: %.1f ms" % (batch, lanes, best * 1e3), flush=True)
