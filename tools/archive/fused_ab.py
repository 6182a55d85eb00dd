#!/usr/bin/env python3
"""Fused vs unfused 2^20-row LDE on one box: bit equality of the outputs, then HIP-event times of every launch of either
sequence on the context's own workspaces (zkhip_ntt_pass which = 2..7) and of the whole LDE both ways.
Usage: fused_ab.py [width=256] [reps=200] [--ab] (--ab: the A/B build with the ZKHIP_FUSED_* knobs)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (tools/archive/ -> the repository root)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
if "--ab" in sys.argv:
    import _ab  # noqa: F401
    sys.argv.remove("--ab")
import numpy as np  # noqa: E402
from zktls_amd.device import Context  # noqa: E402

width = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
check = os.environ.get("FUSED_AB_CHECK", "1") != "0"
hip = C.CDLL("libamdhip64.so")
ctx = Context(0)
log_n = 20
src = ctx.fill_uniform(1, log_n, width)
out = ctx.alloc((width << log_n) * 2)
e0, e1 = C.c_void_p(), C.c_void_p()
hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
st = C.c_void_p(ctx.stream)


def timed(fn, n):
    for _ in range(max(3, n // 10)):
        fn()
    hip.hipEventRecord(e0, st)
    for _ in range(n):
        fn()
    hip.hipEventRecord(e1, st)
    hip.hipEventSynchronize(e1)
    ms = C.c_float()
    hip.hipEventElapsedTime(C.byref(ms), e0, e1)
    return ms.value / n


if check:
    ctx.set_lde_fusion(True)
    ctx.coset_lde(src, log_n, width, out=out)
    a = out.download()
    ctx.set_lde_fusion(False)
    ctx.coset_lde(src, log_n, width, out=out)
    b = out.download()
    same = bool((a == b).all())
    print("fused == unfused:", same, "| nonzero:", int(np.count_nonzero(a[:1 << 20])))
    if not same:
        d = np.flatnonzero(a != b)
        print("first mismatches at", d[:8], "count", d.size)
        sys.exit(1)
    del a, b
res = {}
for which in (2, 3, 4, 5, 6, 7):
    res[which] = timed(lambda w=which: ctx.ntt_pass(src, None, log_n, width, w), reps)
names = {2: "I1 strided->strided", 3: "I2 contiguous", 4: "F1 block->strided", 5: "F2 contiguous", 6: "I1 strided->blocks", 7: "fused I2+F1+F1"}
for w in (2, 3, 4, 5, 6, 7):
    gb = (12 if w == 7 else 8) * (width << log_n) / 1e9
    print("which %d %-22s %.4f ms  %.0f GB/s  %.3f of 8 TB/s" % (w, names[w], res[w], gb / res[w] * 1e3, gb / res[w] * 1e3 / 8000))
ctx.set_lde_fusion(True)
lf = timed(lambda: ctx.coset_lde(src, log_n, width, out=out), max(20, reps // 5))
ctx.set_lde_fusion(False)
lu = timed(lambda: ctx.coset_lde(src, log_n, width, out=out), max(20, reps // 5))
print("LDE fused %.4f ms | unfused %.4f ms | sum of fused launches %.4f | sum of unfused launches %.4f" %
      (lf, lu, res[6] + res[7] + 2 * res[5], res[2] + res[3] + 2 * res[4] + 2 * res[5]))
ctx.close()
