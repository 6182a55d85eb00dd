#!/usr/bin/env python3
"""does proving the shards in flight PHASE-ALIGNED keep the shader clock up in the leaf-hash phase?

The shader clock is sampled as a time line by tools/clockprobe (a one-wave kernel on a high-priority stream every 500 us:
delta s_memtime / delta s_memrealtime x 100 MHz) beside each schedule; every schedule is timed with the profiler absent.

  part 1  operators only (one stream, the headline shapes: coset LDE of 2^20 x 256, leaf hash of 2^21 x 256):
          hash only, LDE only, [LDE, hash] x 8, [LDE x 4, hash x 4] x 2, [LDE x 8, hash x 8]
  part 2  whole proofs of the headline shard, 32 distinct shards per call of zkhip_prove_shards_multi:
          one context + stream per worker (1, 2, 4, 8 in flight), then the lock-step lanes of csrc/batch.h at this shape
          (A/B build: ZKHIP_LOCKSTEP_MAX_CELLS lifts the small-proof bound) -- merged launches of 2 / 4 / 8 members, 1 - 3 lanes
usage: python tools/phase_align_probe.py [n_shards=32]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.dirname(os.path.abspath(__file__))]
os.environ.setdefault("ZKHIP_LOCKSTEP_MAX_CELLS", str(1 << 30))
import _ab  # noqa: F401,E402
from zktls_amd._lib import Params  # noqa: E402
from zktls_amd.device import Context, prove_shards_multi, set_lockstep, lockstep_stats  # noqa: E402

probe = C.CDLL(os.path.join(ROOT, "tools", "clockprobe", "libclockprobe.so"))
probe.clockprobe_start.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
probe.clockprobe_stop.argtypes = [C.c_void_p, C.c_uint32]
probe.clockprobe_stop.restype = C.c_uint32
CAP = 1 << 16


def with_clock(label, fn, unit=None, units=1):
    """runs fn() with the clock probe beside it; prints wall time and the clock's distribution"""
    assert probe.clockprobe_start(0, CAP, 500) == 0
    t0 = time.perf_counter()
    out = fn()
    dt = time.perf_counter() - t0
    buf = np.zeros(CAP * 3, dtype=np.uint64)
    n = probe.clockprobe_stop(buf.ctypes.data, CAP)
    s = buf[: n * 3].reshape(n, 3)
    s = s[s[:, 2] > 0]
    mhz = s[:, 1].astype(np.float64) / s[:, 2].astype(np.float64) * 100.0
    if len(mhz) == 0:
        mhz = np.zeros(1)
    q = np.percentile(mhz, [10, 50, 90])
    per = " = %7.3f ms per %s" % (dt * 1e3 / units, unit) if unit else ""
    print("%-64s %8.2f ms%s | clock MHz mean %4.0f p10 %4.0f p50 %4.0f p90 %4.0f (%d samples)" % (label, dt * 1e3, per, mhz.mean(), q[0], q[1], q[2], len(mhz)), flush=True)
    return out, dt, mhz, s


def series(label, mhz, s, step_ms=2.0):
    """the clock as a coarse time line: mean per `step_ms` of the device's real-time counter"""
    if len(mhz) < 2:
        return
    t = (s[:, 0] - s[0, 0]).astype(np.float64) / 1e5          # ms (100 MHz ticks)
    bins = (t / step_ms).astype(int)
    line = []
    for b in range(bins.max() + 1):
        m = mhz[bins == b]
        line.append("%4.0f" % m.mean() if len(m) else "   .")
    print("    %s, clock per %.0f ms: %s" % (label, step_ms, " ".join(line[:80])), flush=True)


n_shards = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = Context(0)
log_n, w = 20, 256
prm = Params(1, 100, 16)

# ---- part 1: operators
src = ctx.fill_uniform(1, log_n, w)
lde = ctx.alloc((w << log_n) * 2)
dig = ctx.alloc(8 << (log_n + 1))


def op_lde():
    ctx.coset_lde(src, log_n, w, out=lde)


def op_hash():
    ctx.hash_rows([(lde, w)], 2 << log_n, out=dig)


def sched(pattern, reps):
    def run():
        for _ in range(reps):
            for nl, nh in pattern:
                for _ in range(nl):
                    op_lde()
                for _ in range(nh):
                    op_hash()
        ctx.sync()
    return run


for _ in range(3):
    op_lde(); op_hash()
ctx.sync()
print("part 1: operators on one stream (16 LDEs + 16 leaf hashes per schedule unless stated)")
time.sleep(0.5)
_, _, m, s = with_clock("hash only x 16 (after 0.5 s idle)", sched([(0, 1)], 16), "hash", 16)
series("hash only", m, s)
time.sleep(0.5)
_, _, m, s = with_clock("LDE only x 16 (after 0.5 s idle)", sched([(1, 0)], 16), "LDE", 16)
series("LDE only", m, s)
for label, pat, reps in (("[LDE, hash] x 16", [(1, 1)], 16), ("[LDE x 2, hash x 2] x 8", [(2, 2)], 8), ("[LDE x 4, hash x 4] x 4", [(4, 4)], 4),
                         ("[LDE x 8, hash x 8] x 2", [(8, 8)], 2), ("[LDE x 16, hash x 16]", [(16, 16)], 1)):
    time.sleep(0.5)
    sched(pat, 1)()
    _, _, m, s = with_clock(label, sched(pat, reps), "pair", 16)
    series(label, m, s, 4.0)

# ---- part 2: whole proofs
trs = [ctx.gen_trace(1, s_, log_n, w) for s_ in range(n_shards)]
pvs = [[1, s_] for s_ in range(n_shards)]
ctx.sync()
print("part 2: %d distinct headline shards per call of zkhip_prove_shards_multi" % n_shards)
ref = None
for inflight in (1, 2, 4, 8):
    set_lockstep(0)
    prove_shards_multi(trs[: max(inflight, 2)], log_n, w, pvs[: max(inflight, 2)], prm, devices=[0], in_flight=inflight)
    time.sleep(0.3)
    out, dt, m, s = with_clock("one stream per worker, %d in flight" % inflight, lambda: prove_shards_multi(trs, log_n, w, pvs, prm, devices=[0], in_flight=inflight), "shard", n_shards)
    series("%d in flight" % inflight, m, s, 4.0)
    if ref is None:
        ref = [p.tobytes() for p in out]
    else:
        assert [p.tobytes() for p in out] == ref, "bytes differ"
for batch, lanes in ((4, 1), (4, 2), (8, 1), (8, 2), (2, 2), (2, 4), (4, 3), (16, 1), (16, 2)):
    set_lockstep(batch, lanes)
    try:
        prove_shards_multi(trs[: batch * lanes], log_n, w, pvs[: batch * lanes], prm, devices=[0])
        time.sleep(0.3)
        s0 = lockstep_stats()
        out, dt, m, s = with_clock("lock-step, %d members x %d lanes" % (batch, lanes), lambda: prove_shards_multi(trs, log_n, w, pvs, prm, devices=[0]), "shard", n_shards)
        s1 = lockstep_stats()
        series("lock-step %d x %d" % (batch, lanes), m, s, 4.0)
        print("    merged launches %d for %d member requests, %d mixed rendezvous; bytes %s" % (s1[0] - s0[0], s1[1] - s0[1], s1[2] - s0[2],
              "unchanged" if [p.tobytes() for p in out] == ref else "DIFFER"), flush=True)
    except Exception as e:                                     # a shape the lanes cannot hold: say so and go on
        print("lock-step %d x %d: %s" % (batch, lanes, e), flush=True)
set_lockstep(16, 6)
